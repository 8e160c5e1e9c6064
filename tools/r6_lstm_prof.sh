#!/bin/bash
# kernel trace of the train step with the recipes' LSTM prediction network (dec_type=rnn): what the recurrence costs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6lstm; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp; rm -rf /tmp/prof_ts; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_ts -o ts -- python $GRAFT_REPO_ROOT/bench.py --workload train_step --pred-net rnn --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg > $GRAFT_REPO_ROOT/$O/prof_ts.log 2>&1)
db=$(find /tmp/prof_ts -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db --top 120 > $O/lstm_train_step_kernel_stats.csv
python tools/step_sequence.py $db > $O/lstm_train_step_sequence.txt
tail -3 $O/lstm_train_step_sequence.txt
grep -i "lstm\|miopen\|Cijk\|fillBuffer" $O/lstm_train_step_kernel_stats.csv | cut -c1-160 | head -20
grep -o '"ms_per_step": [0-9.]*' $O/prof_ts.log | head -3
