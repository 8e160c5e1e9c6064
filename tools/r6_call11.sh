#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c11; mkdir -p $O
timeout 900 python -m pytest tests/test_joint_gpu.py tests/test_gemm_gpu.py tests/test_model_full.py tests/test_train_step_gpu.py -x -q -m gpu 2>&1 | tail -4
python tools/fc2_epi_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/fc2.txt
timeout 600 python bench.py --workload train_step --steps 10 --warmup 4 --no-cpu-baseline --no-fp32-leg > $O/train_step.json 2> $O/train_step.err; grep -o '"ms_per_step": [0-9.]*' $O/train_step.json | head -3
