#!/bin/bash
# the recipes' own per-GPU batch (egs/train_transducer_bmuf_otfaug.sh:27: batch_size=8) with the LSTM prediction network:
# the train step with the persistent recurrence and with the library's
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for p in 1 0; do
LSTM_ON=$p python - <<'PY'
import os, sys, io, json, contextlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pika_amd", "dropin"))
import pika_amd
from pika_amd.model import lstm
lstm.PERSISTENT = os.environ["LSTM_ON"] == "1"
sys.argv = ["bench.py", "--workload", "train_step", "--pred-net", "rnn", "--batch", "8", "--steps", "20", "--warmup", "4", "--no-cpu-baseline", "--no-fp32-leg"]
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("persistent" if lstm.PERSISTENT else "library   ", "B=8 train step %.2f ms  %.1f utt/s" % (d["ms_per_step"], d["value"]))
PY
done
