#!/bin/bash
# decode-side check: the decode / LAS / FST / MBR GPU tests, then the decode bench leg (search, results, LAS phases)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_decode_step_gpu.py tests/test_decode_full.py tests/test_decode.py tests/test_las.py tests/test_las_kernels_gpu.py tests/test_las_full.py tests/test_fst.py tests/test_mbr.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --workload decode --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/decode_check.log 2>&1
grep -o '"search_s": [0-9.]*\|"results_s": [0-9.]*\|"ms_per_step": [0-9.]*\|"las_rescoring_s": [0-9.]*\|"top1_equals_the_burst_sequence": "[^"]*"\|\["host prep", [0-9.]*\]' gpurun_out/decode_check.log | head -14
