#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c15; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bmuf.py tests/test_mbr.py tests/test_las_kernels_gpu.py -x -q -m gpu 2>&1 | tail -3
(cd /tmp; rm -rf /tmp/prof_dec; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_dec -o dec -- python $GRAFT_REPO_ROOT/bench.py --workload decode --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_dec.log 2>&1)
python tools/decode_anatomy.py $(find /tmp/prof_dec -name '*_results.db' | head -1) | tee $O/decode_anatomy.txt
