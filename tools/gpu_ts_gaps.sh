#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/prof_ts
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_ts -o ts -- python $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg > $GRAFT_REPO_ROOT/gpurun_out/prof_ts.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/prof_ts -name '*_results.db' | head -1)
python tools/rocpd_gaps.py $db --skip-first 3000 > gpurun_out/ts_gaps.txt
cat gpurun_out/ts_gaps.txt
