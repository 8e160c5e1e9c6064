#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_mbr
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_mbr -o mbr -- python $GRAFT_REPO_ROOT/bench.py --workload mbr_step --batch 8 --beam 4 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_mbr.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/prof_mbr -name '*_results.db' | head -1)
python tools/step_chain.py $db | tee gpurun_out/t_mbr_chain.txt
tail -2 gpurun_out/prof_mbr.log | cut -c1-300
