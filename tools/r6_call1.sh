#!/bin/bash
# round 6, call 1: the T_in = 1000 golden on the device; who issues the device copies; the beam-boundary tie probe; baselines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6c1
mkdir -p $O
timeout 900 python -m pytest tests/test_model_full.py -x -q -m gpu -k "benchmarked_length" -s 2>&1 | tail -15 > $O/long_golden.txt
cat $O/long_golden.txt
timeout 600 python tools/decode_tie_probe.py > $O/tie_probe.txt 2>&1; tail -20 $O/tie_probe.txt
(cd /tmp && rm -rf /tmp/prof_cp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_cp -o cp -- python $GRAFT_REPO_ROOT/bench.py --workload decode --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --fst --las > $GRAFT_REPO_ROOT/$O/prof_cp.log 2>&1)
db=$(find /tmp/prof_cp -name '*_results.db' | head -1)
python tools/copy_census.py $db > $O/copy_census.txt 2>&1; cat $O/copy_census.txt
timeout 600 python bench.py --workload train_step --steps 10 --warmup 4 --no-cpu-baseline > $O/train_step.json 2> $O/train_step.err; tail -c 1500 $O/train_step.json
timeout 600 python bench.py --workload mbr_step --batch 8 --beam 4 --steps 4 --warmup 2 --no-cpu-baseline > $O/mbr.json 2> $O/mbr.err; tail -c 1200 $O/mbr.json
