#!/bin/bash
# kernel-trace summary of the HEADLINE command (rnnt_loss_M1 only) -> gpurun_out/m1_kernel_stats.csv
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/prof_m1
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_m1 -o m1 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train-step --no-decode > $GRAFT_REPO_ROOT/gpurun_out/prof_m1.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/prof_m1 -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db --top 12 > gpurun_out/m1_kernel_stats.csv
cat gpurun_out/m1_kernel_stats.csv | cut -c1-200; grep "^{" gpurun_out/prof_m1.log | cut -c1-900
