#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6fc2; mkdir -p $O
for v in none 4 8 16 28 31; do
  if [ $v = none ]; then export PIKA_HIPCC_EXTRA=""; else export PIKA_HIPCC_EXTRA="-DPP_ABL=$v"; fi
  touch pika_amd/csrc/gemm_glds.hip
  python -m pika_amd.build > $O/build_$v.log 2>&1 || { tail -5 $O/build_$v.log; continue; }
  echo "=== PP_ABL=$v"
  python tools/fc2_epi_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/abl_$v.txt
done
