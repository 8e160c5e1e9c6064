#!/bin/bash
mkdir -p gpurun_out/h && cd /root/repo
export TMPDIR=/tmp
PIKA_LAS_TIMING=1 timeout 400 python bench.py --workload decode --batch 64 --fst --las --steps 2 --warmup 1 --no-cpu-baseline --blank-bias 2.5390625 > gpurun_out/h/dec_full.json 2> gpurun_out/h/dec_full.err
cd /tmp; rm -rf /tmp/prof_dec
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_dec -o dec -- python /root/repo/bench.py --workload decode --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --fst --las --blank-bias 2.5390625 > /root/repo/gpurun_out/h/prof.log 2>&1
cd /root/repo
db=$(find /tmp/prof_dec -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db --top 60 > gpurun_out/h/decode_fst_las_kernel_stats.csv
python tools/rocpd_gaps.py $db > gpurun_out/h/decode_fst_las_gaps.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/h/dec_full.json"))
t=d["config"]["timing"]
print({k:v for k,v in t.items() if k!="las_phases_ms"})
for k,v in t.get("las_phases_ms",{}).items():
    print(k, [(n, round(ms,1)) for n,ms in v])
PY
head -45 gpurun_out/h/decode_fst_las_kernel_stats.csv | cut -c1-150; head -12 gpurun_out/h/decode_fst_las_gaps.txt
