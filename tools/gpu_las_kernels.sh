#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_las
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_las -- python $GRAFT_REPO_ROOT/bench.py --workload decode --batch 64 --fst --las --steps 2 --warmup 1 --no-cpu-baseline --blank-bias 2.54 > $GRAFT_REPO_ROOT/gpurun_out/prof_las.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_las -name '*_kernel_stats.csv' | head -1)
python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('las_','lstm_cell_kernel','dgemm_kernel<32, 4>','dgemm_kernel<64, 4>','blstm')):
        print(n[:70].ljust(70), r['Calls'].rjust(6), '%8.1f us avg'%(float(r['AverageNs'])/1e3), '%7.1f ms'%(float(r['TotalDurationNs'])/1e6))
PY
