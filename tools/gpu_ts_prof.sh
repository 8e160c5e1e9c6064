#!/bin/bash
# kernel-trace summary of the bf16 train step -> gpurun_out/ts_kernel_stats.csv (+ optional tag $1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ts
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_ts -o ts -- python $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg > $GRAFT_REPO_ROOT/gpurun_out/prof_ts.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/prof_ts -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db --top 90 > gpurun_out/ts_kernel_stats${1}.csv
rm -rf gpurun_out/prof_ts
tail -c 600 gpurun_out/prof_ts.log | head -c 300
