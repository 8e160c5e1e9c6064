#!/bin/bash
# the whole GPU suite + the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6full; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r6full/bench_default.json").read().strip().splitlines()[-1])
    print("M1", d["value"], d["ms_per_step"], d["roofline"]["frac"])
    ts = d.get("train_step", {}); print("train", ts.get("ms_per_step"), ts.get("value"), (ts.get("roofline") or {}).get("frac"), ts.get("error"))
    print("  lstm", (ts.get("lstm_prediction_net") or {}).get("ms_per_step"), "bf16", (ts.get("bf16_no_parity") or {}).get("ms_per_step"))
    dc = d.get("decode", {}); print("decode", dc.get("ms_per_step"), dc.get("value"), dc.get("error"))
    t = (dc.get("config") or {}).get("timing") or {}; print("  timing", {k: t.get(k) for k in ("search_s", "steps", "encoder_s")})
    for k in ("fst_las", "pipeline", "full"):
        if k in dc: print("  ", k, str(dc[k])[:300])
    m = d.get("mbr_step", {}); print("mbr", m.get("ms_per_step"), m.get("value"), (m.get("config") or {}).get("training_part_ms"), m.get("error"))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r6full/bench_default.err").read()[-1500:])
PY
