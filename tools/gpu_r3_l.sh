#!/bin/bash
# two ranks on ONE MI355X (gloo for the collective): the N > 1 code path of bench.py -- graphed step per rank, BMUF sync
# every 5 steps with the deferred NaN flag, momentum reset in place -- end to end
mkdir -p gpurun_out/l && cd /root/repo
PIKA_BENCH_DEVICE=0 PIKA_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 2 --batch 16 --no-decode > gpurun_out/l/bench_n2.json 2> gpurun_out/l/bench_n2.err
tail -5 gpurun_out/l/bench_n2.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/l/bench_n2.json"))
print(d["n_gpus"], d["value"])
ts=d["train_step"]
print({k:ts.get(k) for k in ("value","ms_per_step","error")}, ts.get("bmuf"))
print(ts.get("bf16_no_parity"))
PY
timeout 600 python -m pytest tests/test_attn_gpu.py tests/test_mixed_gpu.py tests/test_train_step_gpu.py -m gpu -q 2>&1 | tail -3
timeout 300 python bench.py --workload train_step --precision mixed --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | cut -c1-260
timeout 300 python bench.py --workload train_step --precision bf16 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | cut -c1-260
