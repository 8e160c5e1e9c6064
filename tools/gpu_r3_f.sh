#!/bin/bash
mkdir -p gpurun_out/f && cd /root/repo
timeout 300 python tools/blaslt_ref.py > gpurun_out/f/blaslt.txt 2>&1
PIKA_GEMM_PRECISION=mixed timeout 300 python tools/host_bound.py > gpurun_out/f/host_bound_graph.txt 2>&1
PIKA_GEMM_PRECISION=mixed PIKA_TRAIN_GRAPH=0 timeout 300 python tools/host_bound.py > gpurun_out/f/host_bound_eager.txt 2>&1
timeout 600 bash tools/starved_host.sh > gpurun_out/f/starved.txt 2>&1
cat gpurun_out/f/blaslt.txt | grep hipBLASLt; tail -2 gpurun_out/f/host_bound_graph.txt gpurun_out/f/host_bound_eager.txt; cat gpurun_out/f/starved.txt
