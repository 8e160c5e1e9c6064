#!/bin/bash
# the train step on a starved host -- 16 cores, 8 busy processes competing for them (what eight
# ranks + eight loader threads on one node look like to each other).  Prints ms/step of both launch modes, free and starved.
cd /root/repo
export PIKA_GEMM_PRECISION=mixed
run() { python bench.py --workload train_step --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.2f ms/step' % d['ms_per_step'])"; }
echo "graph, free host:     $(run)"
echo "eager, free host:     $(PIKA_TRAIN_GRAPH=0 run)"
pids=""
for i in 0 1 2 3 4 5 6 7; do taskset -c 0-15 python -c "
import time
t=time.time()
while time.time()-t < 170: sum(range(10000))" & pids="$pids $!"; done
sleep 1
echo "graph, starved host:  $(taskset -c 0-15 bash -c "$(declare -f run); run")"
echo "eager, starved host:  $(PIKA_TRAIN_GRAPH=0 taskset -c 0-15 bash -c "$(declare -f run); run")"
kill $pids 2>/dev/null
wait 2>/dev/null
