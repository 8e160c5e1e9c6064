#!/bin/bash
# per-kernel times of the LAS rescoring leg of configs[4] (the search step no longer uses dgemm_kernel, so its rows are LAS's)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_las
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_las -- python $GRAFT_REPO_ROOT/bench.py --workload decode --batch 64 --fst --las --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_las.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_las -name '*_kernel_stats.csv' | head -1)
python - $f <<'PY' | tee gpurun_out/las_kernels.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('las_','lstm_cell_kernel','dgemm_kernel','dgemm_wide','dgemm_sk','blstm','gemm_pp','gemm_nt','softmax','SoftMax','index','split_terms','copyBuffer')):
        print(n[:90].ljust(90), r['Calls'].rjust(6), '%8.1f us avg'%(float(r['AverageNs'])/1e3), '%7.1f ms'%(float(r['TotalDurationNs'])/1e6), 'min %.1f max %.1f'%(float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
tail -c 600 gpurun_out/prof_las.log
