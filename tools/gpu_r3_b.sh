#!/bin/bash
mkdir -p gpurun_out/b && cd /root/repo
timeout 600 python -m pytest tests/test_model_full.py tests/test_model.py tests/test_mixed_gpu.py tests/test_train_step_gpu.py -m gpu -q -s 2>&1 > gpurun_out/b/t_model.log
grep -n "^\[" gpurun_out/b/t_model.log; tail -3 gpurun_out/b/t_model.log
