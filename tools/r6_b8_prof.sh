#!/bin/bash
# kernel totals of the train step at the recipes' own operating point (8 utterances per GPU, V = 6268, LSTM prediction network)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6b8; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp; rm -rf /tmp/prof_ts; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_ts -o ts -- python $GRAFT_REPO_ROOT/bench.py --workload train_step --pred-net rnn --batch 8 --vocab 6268 --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg > $GRAFT_REPO_ROOT/$O/prof_ts.log 2>&1)
db=$(find /tmp/prof_ts -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db --top 60 > $O/b8_train_step_kernel_stats.csv
python tools/step_sequence.py $db > $O/b8_train_step_sequence.txt
tail -2 $O/b8_train_step_sequence.txt
python - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/r6b8/b8_train_step_kernel_stats.csv')))[1:]
tot=0
for r in rows[:28]:
    try:
        print(r[0].replace('(anonymous namespace)::','')[:84].ljust(84), r[1].rjust(6), '%8.1f us'%(float(r[3])/1e3), '%7.2f ms/step'%(float(r[2])/1e6/13))
    except Exception: pass
PY
grep -o '"ms_per_step": [0-9.]*' $O/prof_ts.log | head -2
