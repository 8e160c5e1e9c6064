#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python bench.py --workload decode --steps 1 --warmup 1 --batch ${1:-64} --pred-net ${2:-transformer} > gpurun_out/decode.log 2>&1; echo "rc=$?" >> gpurun_out/decode.log
tail -4 gpurun_out/decode.log
