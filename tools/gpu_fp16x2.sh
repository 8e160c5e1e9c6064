#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mixed_gpu.py tests/test_decode_full.py tests/test_decode.py tests/test_decode_step_gpu.py -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --workload decode --batch 64 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['config']['timing']; print('decode', d['value'], d['ms_per_step'], t['search_s'], d['config']['labels_per_utt_top1'])
"
