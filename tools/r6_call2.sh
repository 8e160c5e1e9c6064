#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6c2
mkdir -p $O
timeout 900 python -m pytest tests/test_model_full.py -q -m gpu -k "benchmarked_length" -s 2>&1 | tail -12 > $O/long_golden.txt
cat $O/long_golden.txt
timeout 900 python -m pytest tests/test_mbr.py -x -q -m gpu -s 2>&1 | tail -25 > $O/test_mbr.txt; cat $O/test_mbr.txt
timeout 600 python bench.py --workload mbr_step --batch 8 --beam 4 --steps 6 --warmup 3 --no-cpu-baseline > $O/mbr.json 2> $O/mbr.err; tail -c 900 $O/mbr.json; tail -5 $O/mbr.err
PIKA_TRAIN_GRAPH=0 timeout 600 python bench.py --workload mbr_step --batch 8 --beam 4 --steps 6 --warmup 3 --no-cpu-baseline > $O/mbr_eager.json 2> $O/mbr_eager.err; tail -c 400 $O/mbr_eager.json
