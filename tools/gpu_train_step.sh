#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py --workload train_step --steps 3 --warmup 1 --batch ${1:-32} > gpurun_out/train_step.log 2>&1; echo "rc=$?" >> gpurun_out/train_step.log
tail -5 gpurun_out/train_step.log
export TMPDIR=/tmp; cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ts; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ts -- python $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 2 --warmup 1 --batch ${1:-32} > $GRAFT_REPO_ROOT/gpurun_out/prof_ts.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_ts/**/*_kernel_stats.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:40]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(5), '%9.1f us avg'%(float(r['AverageNs'])/1e3), '%6.2f%%'%(100*float(r['TotalDurationNs'])/tot))
PY
