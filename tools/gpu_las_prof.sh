#!/bin/bash
# kernel trace of the full configs[4] decode (FST fusion + LAS rescoring): the LAS part of the last batch as a timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/prof_las
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_las -o las -- python $GRAFT_REPO_ROOT/bench.py --workload decode --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --fst --las > $GRAFT_REPO_ROOT/gpurun_out/prof_las.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/prof_las -name '*_results.db' | head -1)
python tools/las_timeline.py $db | tee gpurun_out/las_timeline.txt
tail -c 300 gpurun_out/prof_las.log
