#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 256 384 512 768 1024; do
  echo "== split target $t"; PIKA_GEMM_SPLIT_TARGET=$t timeout 600 python bench.py --workload train_step --steps 4 --warmup 2 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
