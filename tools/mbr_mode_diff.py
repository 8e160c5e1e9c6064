"""Debug: per-parameter gradient difference of the native MBR step (tests/test_mbr.py fixture) between the exact mode and
another arithmetic mode, split into the RNN-T part and the risk part.  python tools/mbr_mode_diff.py [mode]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_mbr as T  # noqa: E402
from pika_amd import gemm as G  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "mixed"
dev = "cuda:0"
res = {}
for prec in ("fp32", mode):
    G.PRECISION = prec
    for part in ("rnnt", "risk", "both"):
        os.environ["PIKA_MBR_DEBUG_PART"] = part
        M, got, want = T._native_step_vs_script_golden(dev, _raw=True)
        res[(prec, part)] = got
names = res[("fp32", "both")]["names"]
for part in ("rnnt", "risk", "both"):
    a, b = res[("fp32", part)]["grads"], res[(mode, part)]["grads"]
    rows = []
    for n, x, y in zip(names, a, b):
        d = float(np.abs(x - y).max())
        rows.append((d / (float(np.abs(x).max()) + 1e-30), n, float(np.abs(x).max()), float(np.abs(y).max())))
    rows.sort(reverse=True)
    print("== part %s: worst parameters (|diff|max / |g|max, name, max|g| exact, max|g| %s)" % (part, mode))
    for r in rows[:12]:
        print("   %.3e  %-50s %.3e %.3e" % r)
