#!/bin/bash
mkdir -p gpurun_out/e && cd /root/repo
timeout 600 python tools/pp_bench.py 256 > gpurun_out/e/pp_bench.txt 2>&1
PIKA_GEMM_PP_STG=1 timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_mixed_gpu.py tests/test_joint_gpu.py tests/test_model_full.py -m gpu -q -x > gpurun_out/e/t_stg.log 2>&1
cat gpurun_out/e/pp_bench.txt; tail -5 gpurun_out/e/t_stg.log
