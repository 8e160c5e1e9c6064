#!/bin/bash
# round-end check: full GPU suite, smoke(), the default bench line
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/final_pytest_full.txt 2>&1; grep -E "passed|failed|FAILED|ERROR" gpurun_out/final_pytest_full.txt | tee gpurun_out/final_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/gpu_default_bench.sh
