#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_decode_step_gpu.py tests/test_decode_full.py tests/test_decode.py tests/test_las.py tests/test_las_kernels_gpu.py tests/test_mbr.py -q -m gpu 2>&1 | tail -8

PIKA_LAS_TIMING=1 timeout 600 python bench.py --workload decode --batch 64 --fst --las --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['config']['timing']; print('pipeline', d['value'], d['ms_per_step'], t['search_s'], t['terms'], t['las_s'], t.get('las_row_steps'), [(a,round(b,1)) for a,b in t['las_phases_ms']['fw']])
"
