#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_rnnt_loss_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --workload rnnt_loss_M1p --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('M1p', d['ms_per_step'], 'fwd', r['forward_ms'], 'bwd', r['backward_ms'], 'frac', r['frac'])"; done
