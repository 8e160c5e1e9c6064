#!/bin/bash
# Kernel trace of bench.py --workload decode (configs[4] shape, graph replays traced per kernel) and the anatomy of the
# encoder pass between two searches: tools/decode_anatomy.py on the rocpd database.   -> gpurun_out/r6_decode_anatomy.txt
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_dec
PIKA_BENCH_WATCHDOG=600 timeout 800 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_dec -- python $R/bench.py --workload decode --steps 3 --warmup 1 --batch 64 --no-cpu-baseline ${DECODE_ARGS} > $R/gpurun_out/prof_dec.log 2>&1
cd $R
tail -1 gpurun_out/prof_dec.log | cut -c1-600
DB=$(find /tmp/prof_dec -name "*_results.db" | head -1)
python tools/decode_anatomy.py $DB | tee gpurun_out/r6_decode_anatomy.txt
python tools/decode_anatomy.py $DB --seq > gpurun_out/r6_decode_anatomy_seq.txt
