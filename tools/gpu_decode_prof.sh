#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_dec
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec -- python $R/bench.py --workload decode --steps 1 --warmup 0 --batch 64 --pred-net ${1:-rnn} > $R/gpurun_out/prof_dec.log 2>&1
cd $R
python - <<'PY'
import csv,glob,os
f=max(glob.glob('gpurun_out/prof_dec/**/*_kernel_stats.csv',recursive=True), key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:28]:
    print(r['Name'][:95].ljust(95), r['Calls'].rjust(6), '%8.1f us avg'%(float(r['AverageNs'])/1e3), '%7.1f ms'%(float(r['TotalDurationNs'])/1e6))
PY
rm -rf gpurun_out/prof_dec
