#!/bin/bash
# configs[4] in full (FST-fused search + fw/bw LAS rescoring) with the rescoring phases
cd /root/repo; mkdir -p gpurun_out
timeout 600 python bench.py --workload decode --batch 64 --fst --las --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['config']['timing']; print('pipeline', d['value'], d['ms_per_step'], t['search_s'], t['las_s'], {k:[(a[:10],round(b,1)) for a,b in v] for k,v in t.get('las_phases_ms',{}).items()})
"
