#!/bin/bash
# the default bench line only (what the driver runs), with the decode pipeline leg's LAS phases
cd /root/repo; mkdir -p gpurun_out
timeout 1200 python bench.py > gpurun_out/default_bench.json 2> gpurun_out/default_bench.err; tail -2 gpurun_out/default_bench.err | cut -c1-200
python - <<'PY'
import json
for l in open('gpurun_out/default_bench.json'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'])
        ts=d.get('train_step',{}); print('train', {k:ts.get(k) for k in ('value','ms_per_step')}, ts.get('roofline',{}).get('frac'))
        dc=d.get('decode',{}); print('decode', dc.get('value'), dc.get('ms_per_step'), dc.get('exact_step_products'))
        w=dc.get('with_fst_and_las',{}); print('pipeline', {k:w.get(k) for k in ('value','ms_per_step','search_s','las_rescoring_s','las_phases_ms')})
PY
