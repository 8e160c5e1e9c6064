#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_decode_full.py tests/test_decode_step_gpu.py tests/test_las.py tests/test_las_kernels_gpu.py -x -q -m gpu 2>&1 | tail -5
for prec in bf16x3 fp32; do
PIKA_DECODE_PRECISION=$prec PIKA_LAS_TIMING=1 timeout 600 python bench.py --workload decode --batch 64 --fst --las --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/s_decode_$prec.json 2> gpurun_out/s_decode_$prec.err; tail -2 gpurun_out/s_decode_$prec.err
python - $prec <<'PY'
import json, sys
for l in open('gpurun_out/s_decode_%s.json' % sys.argv[1]):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); dd=d.get('decode', d)
        t=dd['config'].get('timing')
        print(sys.argv[1], dd.get('value'), dd.get('ms_per_step'), {k:t[k] for k in ('search_s','results_s','steps','terms','las_s')}, dd['config'].get('labels_per_utt_top1'), [ (a, round(b,1)) for a,b in t.get('las_phases_ms',{}).get('fw',[])])
PY
done
