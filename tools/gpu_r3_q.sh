#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_las_kernels_gpu.py -x -q -m gpu -k "blstm" 2>&1 | tail -15
timeout 120 python tools/blstm_bench.py 2>&1 | tail -4
