#!/bin/bash
mkdir -p gpurun_out/i && cd /root/repo
timeout 600 python bench.py --workload mbr_step --batch 8 --beam 4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/i/mbr.json 2> gpurun_out/i/mbr.err
PIKA_LAS_TIMING=1 timeout 400 python bench.py --workload decode --batch 64 --fst --las --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/i/dec_full.json 2> gpurun_out/i/dec_full.err
timeout 300 python -m pytest tests/test_mbr.py -m gpu -q > gpurun_out/i/t_mbr.log 2>&1
cut -c1-700 gpurun_out/i/mbr.json; tail -3 gpurun_out/i/mbr.err; tail -2 gpurun_out/i/t_mbr.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/i/dec_full.json"))
t=d["config"]["timing"]
print(d["value"], d["ms_per_step"], d["config"].get("labels_per_utt_top1_quartiles"), {k:v for k,v in t.items() if k!="las_phases_ms"})
for k,v in t.get("las_phases_ms",{}).items():
    print(k, [(n, round(ms,1)) for n,ms in v])
PY
