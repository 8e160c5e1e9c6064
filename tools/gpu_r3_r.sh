#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_mixed_gpu.py tests/test_las.py -x -q -m gpu 2>&1 | tail -5
PIKA_LAS_TIMING=1 timeout 600 python bench.py --workload decode --batch 64 --fst --las --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r_decode.json 2> gpurun_out/r_decode.err; tail -3 gpurun_out/r_decode.err
python - <<'PY'
import json
for l in open('gpurun_out/r_decode.json'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); dd=d.get('decode', d)
        print(json.dumps({k:dd[k] for k in dd if k in ('value','ms_per_step')}), json.dumps(dd.get('with_fst_and_las')), json.dumps(dd['config'].get('timing')))
PY
