#!/bin/bash
mkdir -p gpurun_out/m && cd /root/repo
timeout 600 python -m pytest tests/test_attn_gpu.py tests/test_mixed_gpu.py tests/test_model_full.py tests/test_model.py -m gpu -q 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_ts
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_ts -o ts -- python /root/repo/bench.py --workload train_step --precision mixed --steps 10 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/m/prof_ts.log 2>&1
cd /root/repo
db=$(find /tmp/prof_ts -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db --top 110 > gpurun_out/m/train_step_mixed_graph_kernel_stats.csv
grep -i "attn\|keep_bits\|mask_bits" gpurun_out/m/train_step_mixed_graph_kernel_stats.csv | cut -c1-140
tail -1 gpurun_out/m/train_step_mixed_graph_kernel_stats.csv
timeout 300 python bench.py --workload train_step --precision mixed --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | cut -c1-230
timeout 300 python bench.py --workload train_step --precision bf16 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | cut -c1-230
