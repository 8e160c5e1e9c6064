#!/bin/bash
# kernel-trace summary of the full configs[4] decode (FST fusion + LAS rescoring) -> gpurun_out/las_kernel_stats.csv
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/prof_las
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_las -o las -- python $GRAFT_REPO_ROOT/bench.py --workload decode --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --fst --las --blank-bias ${1:-0.5859375} > $GRAFT_REPO_ROOT/gpurun_out/prof_las.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/prof_las -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db --top 60 > gpurun_out/las_kernel_stats.csv
python tools/rocpd_gaps.py $db > gpurun_out/las_gaps.txt
head -30 gpurun_out/las_kernel_stats.csv | cut -c1-160; head -12 gpurun_out/las_gaps.txt; tail -c 400 gpurun_out/prof_las.log
