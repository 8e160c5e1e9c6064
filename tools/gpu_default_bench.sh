#!/bin/bash
# the default `python bench.py` line of the tree -> gpurun_out/bench_default.json (copy to profiles/rN_bench_default_run.json)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("M1", d["value"], d["ms_per_step"], d["roofline"]["frac"])
print("M1p", d["rnnt_loss_M1p"]["ms_per_step"], d["rnnt_loss_M1p"]["roofline"]["frac"])
ts = d["train_step"]; print("train", ts["ms_per_step"], ts["roofline"]["frac"], "lstm", ts["lstm_prediction_net"]["ms_per_step"])
dc = d["decode"]; print("decode", dc["ms_per_step"], dc["config"]["timing"]["search_s"], "full", dc["with_fst_and_las"]["ms_per_step"], dc["with_fst_and_las"]["las_rescoring_s"])
m = d["mbr_step"]; print("mbr", m["ms_per_step"], m["value"])
PY
