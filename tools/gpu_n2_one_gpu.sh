#!/bin/bash
# two ranks on ONE GPU over gloo: the N > 1 code path of bench.py (BMUF sync every 5 steps, graphed step, device-resident
# loader batches, decode replicas) end to end where no multi-GPU box is available
cd /root/repo; mkdir -p gpurun_out
PIKA_BENCH_DEVICE=0 PIKA_BENCH_BACKEND=gloo PIKA_BENCH_WATCHDOG=900 timeout 1000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/n2.json 2> gpurun_out/n2.err
tail -3 gpurun_out/n2.err | cut -c1-300
python - <<'PY'
import json
for l in open('gpurun_out/n2.json'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        print(d['n_gpus'], d['value'], d['ms_per_step'])
        ts=d.get('train_step',{}); print('train', {k:ts.get(k) for k in ('value','ms_per_step','error')}, ts.get('bmuf'))
        dc=d.get('decode',{}); print('decode', dc.get('value'), dc.get('ms_per_step'), dc.get('error'))
PY
