#!/bin/bash
# round-end check: full GPU suite, smoke(), a 200-step soak of the graphed train step, the default bench line
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/final_pytest_full.txt 2>&1; grep -E "passed|failed|FAILED|ERROR" gpurun_out/final_pytest_full.txt | tee gpurun_out/final_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --workload train_step --steps 200 --warmup 5 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('soak 200 steps', d['value'], d['ms_per_step'], d['roofline']['frac'])
"
bash tools/gpu_default_bench.sh
