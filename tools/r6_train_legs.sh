#!/bin/bash
# the train-step legs of the default bench, three times in a row: do the secondary legs (LSTM prediction net, bf16) repeat?
mkdir -p gpurun_out/r6legs
for i in 1 2 3; do
  timeout 900 python bench.py --no-m1-variants --no-decode --no-mbr --no-fp32-leg --no-cpu-baseline > gpurun_out/r6legs/run$i.json 2> gpurun_out/r6legs/run$i.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r6legs/run$i.json"))["train_step"]
print("run $i", round(d["ms_per_step"],2), d["config"].get("graphs"), d["config"].get("ms_between_step_calls"))
for k in ("lstm_prediction_net","bf16_no_parity"):
    if k in d: print("   ", k, round(d[k]["ms_per_step"],2), d[k].get("graphs"), d[k].get("ms_between_step_calls"))
PY
done
