#!/bin/bash
mkdir -p gpurun_out/c && cd /root/repo
timeout 600 python -m pytest tests/test_train_step_gpu.py -m gpu -q -s -k "graphed" > gpurun_out/c/t.log 2>&1
grep -n "largest\|Error\|assert" gpurun_out/c/t.log | cut -c1-600 | head -20
