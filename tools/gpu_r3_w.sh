#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/w_pytest.txt 2>&1
grep -n "FAILED\|passed\|failed" gpurun_out/w_pytest.txt | head -20
grep -n "Error\|assert " gpurun_out/w_pytest.txt | head -30
