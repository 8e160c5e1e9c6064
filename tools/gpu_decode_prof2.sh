#!/bin/bash
# kernel profile of ONE batch decode (no calibration decodes): bash tools/gpu_decode_prof2.sh <blank_bias> [pred_net]
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_dec
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec -- python $R/bench.py --workload decode --steps 1 --warmup 0 --batch 64 --pred-net ${2:-transformer} --blank-bias $1 > $R/gpurun_out/prof_dec.log 2>&1
cd $R
tail -1 gpurun_out/prof_dec.log | cut -c1-400
python - <<'PY'
import csv,glob,os
f=max(glob.glob('gpurun_out/prof_dec/**/*_kernel_stats.csv',recursive=True), key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6, "kernels", sum(int(r['Calls']) for r in rows))
for r in rows[:38]:
    print(r['Name'].replace('(anonymous namespace)::','').replace('at::native::','')[:100].ljust(100), r['Calls'].rjust(6), '%8.1f us avg'%(float(r['AverageNs'])/1e3), '%7.1f ms'%(float(r['TotalDurationNs'])/1e6))
PY
rm -rf gpurun_out/prof_dec
