#!/bin/bash
# full GPU suite + the default bench line (what the driver runs at round end)
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 > gpurun_out/v_pytest.txt; cat gpurun_out/v_pytest.txt
timeout 1200 python bench.py > gpurun_out/v_bench.json 2> gpurun_out/v_bench.err; tail -3 gpurun_out/v_bench.err | cut -c1-300
python - <<'PY'
import json
for l in open('gpurun_out/v_bench.json'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])
        ts=d.get('train_step',{}); print('train', {k:ts.get(k) for k in ('value','ms_per_step')}, ts.get('roofline',{}).get('frac'))
        dc=d.get('decode',{}); print('decode', dc.get('value'), dc.get('ms_per_step'), dc.get('two_term_mode'), dc.get('with_fst_and_las'))
PY
