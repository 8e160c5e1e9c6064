#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_decode_step_gpu.py tests/test_decode_full.py tests/test_las_kernels_gpu.py -x -q -m gpu 2>&1 | tail -4
for prec in fp32 bf16x3; do
PIKA_DECODE_PRECISION=$prec timeout 600 python bench.py --workload decode --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/u_decode_$prec.json 2> gpurun_out/u_decode_$prec.err; tail -2 gpurun_out/u_decode_$prec.err | cut -c1-200
python - $prec <<'PY'
import json, sys
for l in open('gpurun_out/u_decode_%s.json' % sys.argv[1]):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); dd=d.get('decode', d)
        t=dd['config'].get('timing')
        print(sys.argv[1], dd.get('value'), dd.get('ms_per_step'), {k:t[k] for k in ('search_s','results_s','steps','terms')}, dd['config'].get('labels_per_utt_top1'))
PY
done
timeout 600 python bench.py --workload mbr_step --batch 8 --beam 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-400
