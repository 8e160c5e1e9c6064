#!/bin/bash
# kernel trace of the train step: per-kernel summary (profiles/r6_train_step_kernel_stats.csv) + the launch sequence of one step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6ts; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp; rm -rf /tmp/prof_ts; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_ts -o ts -- python $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg > $GRAFT_REPO_ROOT/$O/prof_ts.log 2>&1)
db=$(find /tmp/prof_ts -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db --top 90 > $O/train_step_kernel_stats.csv
python tools/step_sequence.py $db > $O/train_step_sequence.txt
tail -3 $O/train_step_sequence.txt; head -12 $O/train_step_kernel_stats.csv | cut -c1-120
grep -o '"ms_per_step": [0-9.]*' $O/prof_ts.log | head -3
