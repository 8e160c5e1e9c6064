#!/bin/bash
mkdir -p gpurun_out/d && cd /root/repo
timeout 900 python -m pytest tests/test_decode_full.py tests/test_decode.py tests/test_decode_step_gpu.py tests/test_fst.py tests/test_las.py tests/test_las_kernels_gpu.py tests/test_loader.py tests/test_frontend.py -m gpu -q -s > gpurun_out/d/t.log 2>&1
timeout 300 python bench.py --workload decode --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/d/dec_tr.json 2> gpurun_out/d/dec_tr.err
timeout 300 python bench.py --workload decode --pred-net rnn --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/d/dec_rnn.json 2> gpurun_out/d/dec_rnn.err
timeout 400 python bench.py --workload decode --fst --las --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/d/dec_full.json 2> gpurun_out/d/dec_full.err
grep -n "passed\|failed\|FAILED\|mode:" gpurun_out/d/t.log | cut -c1-300 | tail -15
for f in gpurun_out/d/dec_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print({k:d.get(k) for k in ("value","ms_per_step")}, {k:v for k,v in d.get("config",{}).items() if k!="workload"})
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-500:])
PY
done
