#!/bin/bash
# What bounds a K-tile of gemm_pp?  Timing-only ablation builds (wrong results): no fragment reads behind the first K-tile,
# no LDS-DMA pieces behind the prologue, neither (MFMA segments + barriers alone).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6abl; mkdir -p $O
for v in none 1 2 3; do
  if [ $v = none ]; then export PIKA_HIPCC_EXTRA=""; else export PIKA_HIPCC_EXTRA="-DPP_ABL=$v"; fi
  touch pika_amd/csrc/gemm_glds.hip
  python -m pika_amd.build > $O/build_$v.log 2>&1 || { tail -5 $O/build_$v.log; continue; }
  echo "=== PP_ABL=$v"
  python tools/pp_bench.py 256 2>&1 | grep -v amdgpu.ids | tee $O/abl_$v.txt
done
