#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
PIKA_BENCH_DEVICE=0 PIKA_BENCH_BACKEND=gloo PIKA_BENCH_WATCHDOG=900 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r6_n2.json 2> gpurun_out/r6_n2.err
tail -3 gpurun_out/r6_n2.err | cut -c1-300
python - <<'PY'
import json
for l in open('gpurun_out/r6_n2.json'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        print(d['n_gpus'], d['value'], d['ms_per_step'])
        ts=d.get('train_step',{}); print('train', {k:ts.get(k) for k in ('value','ms_per_step','vs_n1','speedup_over_n1','error')}, ts.get('n1_sub_run'), ts.get('bmuf'))
        dc=d.get('decode',{}); print('decode', dc.get('value'), dc.get('ms_per_step'), dc.get('error'))
        m=d.get('mbr_step',{}); print('mbr', m.get('value'), m.get('ms_per_step'), m.get('error'))
        print('summary', json.dumps(d.get('scaling_summary'))[:900])
PY
