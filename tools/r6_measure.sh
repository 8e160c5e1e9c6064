#!/bin/bash
# round-6 measurement refresh: M1 kernel stats (rocprofv3 --stats), PMC traffic of train step / MBR step / M1' / decode step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
(cd /tmp; rm -rf /tmp/prof_m1; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_m1 -o m1 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train-step --no-decode --no-mbr --no-m1-variants > $R/gpurun_out/m1_prof.log 2>&1)
db=$(find /tmp/prof_m1 -name '*_results.db' | head -1); python tools/rocpd_stats.py $db --top 12 > gpurun_out/r6_rnnt_loss_M1_kernel_stats.csv; head -6 gpurun_out/r6_rnnt_loss_M1_kernel_stats.csv | cut -c1-160
grep -o '"roofline": {[^}]*}' gpurun_out/m1_prof.log | head -1 | cut -c1-400
bash tools/r6_pmc.sh train_step 6 --workload train_step --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-leg
bash tools/r6_pmc.sh mbr_step 4 --workload mbr_step --batch 8 --beam 4 --steps 3 --warmup 1 --no-cpu-baseline
bash tools/r6_pmc.sh m1p 6 --workload rnnt_loss_M1p --steps 4 --warmup 2 --no-cpu-baseline
bash tools/gpu_pmc_decode.sh 2>&1 | tail -25
