#!/bin/bash
# host CPU time of the training thread after moving the label/length uploads to the loader's stream
cd /root/repo; mkdir -p gpurun_out
python tools/host_bound.py --profile > gpurun_out/p_host_graph.txt 2>&1
PIKA_TRAIN_GRAPH=0 python tools/host_bound.py > gpurun_out/p_host_eager.txt 2>&1
head -50 gpurun_out/p_host_graph.txt; grep "host enqueue" gpurun_out/p_host_eager.txt
python -m pytest tests/test_loader.py tests/test_train_step_gpu.py -x -q -m gpu 2>&1 | tail -3
