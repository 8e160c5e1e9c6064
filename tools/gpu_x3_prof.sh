#!/bin/bash
# kernel-trace summary of the train step in the bf16x3 arithmetic mode -> gpurun_out/x3_kernel_stats.csv
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_x3
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_x3 -o x3 -- python $GRAFT_REPO_ROOT/bench.py --workload train_step --precision ${1:-bf16x3} --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-leg > $GRAFT_REPO_ROOT/gpurun_out/prof_x3.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/prof_x3 -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db --top 70 > gpurun_out/x3_kernel_stats.csv
rm -rf gpurun_out/prof_x3
head -45 gpurun_out/x3_kernel_stats.csv | cut -c1-150
