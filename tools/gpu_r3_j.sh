#!/bin/bash
mkdir -p gpurun_out/j && cd /root/repo
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/j/bench_default.json 2> gpurun_out/j/bench_default.err ) 2> gpurun_out/j/bench_time.txt
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/j/t_all.log 2>&1
tail -3 gpurun_out/j/bench_time.txt; tail -4 gpurun_out/j/t_all.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/j/bench_default.json"))
print("M1", d["value"], d["ms_per_step"], d["roofline"]["frac"])
ts=d["train_step"]
print("train", {k:ts.get(k) for k in ("value","ms_per_step","dtype")}, ts.get("roofline",{}).get("frac"), ts.get("error"))
for k in ("bf16_no_parity","bf16x3","fp32_split"):
    if k in ts: print(k, ts[k]["ms_per_step"], ts[k].get("roofline_frac"))
print("cpu", ts.get("cpu_baseline"))
dd=d["decode"]
print("decode", dd.get("value"), dd.get("ms_per_step"), dd.get("error"), dd.get("config",{}).get("timing"))
print("pipeline", dd.get("with_fst_and_las"))
PY
