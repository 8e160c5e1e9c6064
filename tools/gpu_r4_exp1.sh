#!/bin/bash
# round-4 experiment call: the two new train-step tests, loss curves per arithmetic mode, the search step's launch chain on
# the speech-like workload with 8 / 16 prefix positions per request batch in dstep_attn_kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/*
timeout 300 python -m pytest tests/test_train_step_gpu.py -x -q -s -k "lazy_log_probs or trains_like" > gpurun_out/exp1_tests.log 2>&1
tail -15 gpurun_out/exp1_tests.log
timeout 300 python tools/curve_modes.py > gpurun_out/exp1_curves.log 2>&1
cat gpurun_out/exp1_curves.log
export TMPDIR=/tmp
for u in 8 16; do
  cd /tmp; rm -rf /tmp/prof_dec
  PIKA_DSTEP_ATTN_UNROLL=$u timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_dec -o dec -- python $GRAFT_REPO_ROOT/bench.py --workload decode --batch 64 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/exp1_dec_u$u.log 2>&1
  cd $GRAFT_REPO_ROOT
  db=$(find /tmp/prof_dec -name '*_results.db' | head -1)
  python tools/step_chain.py $db > gpurun_out/exp1_chain_u$u.txt
  echo "== unroll $u"; cat gpurun_out/exp1_chain_u$u.txt; grep -o '"search_s": [0-9.]*' gpurun_out/exp1_dec_u$u.log | head -3
done
PIKA_DSTEP_ATTN_UNROLL=16 timeout 300 python -m pytest tests/test_decode_step_gpu.py tests/test_decode_full.py -x -q > gpurun_out/exp1_dec_tests_u16.log 2>&1
tail -5 gpurun_out/exp1_dec_tests_u16.log
