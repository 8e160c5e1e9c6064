#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_dec
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_dec -o dec -- python $GRAFT_REPO_ROOT/bench.py --workload decode --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --blank-bias 2.54 > $GRAFT_REPO_ROOT/gpurun_out/prof_dec.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/prof_dec -name '*_results.db' | head -1)
python tools/step_chain.py $db | tee gpurun_out/ab_chain.txt
