#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_las_kernels_gpu.py -x -q -m gpu -k "prefix_sharing or fused_scoring" 2>&1 | grep -v "^$" | tail -40
