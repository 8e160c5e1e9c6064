#!/bin/bash
# round 3, GPU call A: new mixed-mode kernels + full golden + train-step timing + kernel trace
mkdir -p gpurun_out/a && cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mixed_gpu.py -m gpu -q -x 2>&1 | tail -40 > gpurun_out/a/t_mixed.log
timeout 600 python -m pytest tests/test_model_full.py tests/test_model.py tests/test_attn_gpu.py -m gpu -q -s 2>&1 | tail -60 > gpurun_out/a/t_model.log
timeout 300 python bench.py --workload train_step --precision mixed --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/a/b_mixed.json 2> gpurun_out/a/b_mixed.err
timeout 300 python bench.py --workload train_step --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/a/b_bf16.json 2> gpurun_out/a/b_bf16.err
cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/a/prof -o mixed -- python /root/repo/bench.py --workload train_step --precision mixed --steps 10 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/a/prof.log 2>&1
cd /root/repo
db=$(find gpurun_out/a/prof -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db --top 110 > gpurun_out/a/mixed_kernel_stats.csv
rm -rf gpurun_out/a/prof
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > gpurun_out/a/t_all.log
tail -5 gpurun_out/a/t_mixed.log gpurun_out/a/t_model.log gpurun_out/a/t_all.log; cat gpurun_out/a/b_mixed.json | cut -c1-600
