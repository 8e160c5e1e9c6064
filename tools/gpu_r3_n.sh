#!/bin/bash
mkdir -p gpurun_out/n && cd /root/repo
PIKA_GEMM_PRECISION=mixed timeout 900 python -m pytest tests -m gpu -q > gpurun_out/n/t_all_mixed_default.log 2>&1
grep -n "passed\|failed\|FAILED" gpurun_out/n/t_all_mixed_default.log | head -30
