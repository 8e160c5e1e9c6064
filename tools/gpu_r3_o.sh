#!/bin/bash
mkdir -p gpurun_out/o && cd /root/repo
timeout 300 python tools/host_bound.py > gpurun_out/o/host_bound_graph.txt 2>&1
PIKA_TRAIN_GRAPH=0 timeout 300 python tools/host_bound.py > gpurun_out/o/host_bound_eager.txt 2>&1
timeout 300 python bench.py --workload decode --batch 64 --pred-net rnn --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/o/dec_rnn.json 2>/dev/null
timeout 300 python bench.py --workload train_step --precision bf16x3 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/o/ts_x3.json 2>/dev/null
timeout 300 python bench.py --workload rnnt_loss_M1p --steps 10 --warmup 3 > gpurun_out/o/m1p.json 2>/dev/null
timeout 300 python bench.py --workload mbr_step --batch 8 --beam 4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/o/mbr.json 2>/dev/null
grep -h "launch mode\|host enqueue" gpurun_out/o/host_bound_*.txt
python - <<'PY'
import json
for f in ("dec_rnn","ts_x3","m1p","mbr"):
    try:
        d=json.load(open("gpurun_out/o/%s.json"%f))
        print(f, d.get("value"), d.get("ms_per_step"), d.get("roofline",{}).get("frac"), d.get("config",{}).get("timing"))
    except Exception as e: print(f, "ERR", e)
PY
