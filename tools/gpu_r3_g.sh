#!/bin/bash
mkdir -p gpurun_out/g && cd /root/repo
timeout 600 python -m pytest tests/test_las.py tests/test_las_kernels_gpu.py tests/test_decode_step_gpu.py -m gpu -q > gpurun_out/g/t.log 2>&1
timeout 400 python bench.py --workload decode --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/g/dec_tr.json 2> gpurun_out/g/dec_tr.err
timeout 400 python bench.py --workload decode --batch 64 --fst --las --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/g/dec_full.json 2> gpurun_out/g/dec_full.err
PIKA_LAS_GRAPH=0 timeout 400 python bench.py --workload decode --batch 64 --fst --las --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/g/dec_full_nograph.json 2> gpurun_out/g/dec_full_nograph.err
tail -4 gpurun_out/g/t.log
for f in gpurun_out/g/dec_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    c=d.get("config",{})
    print({k:d.get(k) for k in ("value","ms_per_step")}, {k:c.get(k) for k in ("labels_per_utt_top1","labels_per_utt_top1_quartiles","calibration_labels","blank_bias","timing")})
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:], open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
