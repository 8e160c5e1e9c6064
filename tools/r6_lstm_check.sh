#!/bin/bash
# the recipes' LSTM prediction network on the persistent recurrence: reference goldens, graph replays, the train-step leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_model_full.py tests/test_train_step_gpu.py -q -m gpu -k "lstm or rnn" 2>&1 | tail -8
timeout 300 python bench.py --workload train_step --pred-net rnn --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('rnn', r['ms_per_step'], r['value'], r['config'].get('loss'))"
