#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_decode_step_gpu.py -x -q -m gpu -k "dgemm or fc2" 2>&1 | tail -4
timeout 500 python tools/decode_two_term_check.py 2>&1 | tail -4
PIKA_DECODE_PRECISION=fp16x2 timeout 600 python bench.py --workload decode --batch 64 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['config']['timing']; print('fp16x2', d['value'], d['ms_per_step'], t['search_s'], t['terms'], d['config']['labels_per_utt_top1'])
"
