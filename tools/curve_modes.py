"""Loss curves of the script's loop in the arithmetic modes (tests/test_train_step_gpu.py::_loss_curve) for a few learning
rates, as means over one pass of the recurring batches: what the thresholds of test_mixed_arithmetic_trains_like_fp32 were
read from.  GPU box: python tools/curve_modes.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pika_amd  # noqa: E402,F401
import torch  # noqa: E402

import test_train_step_gpu as T  # noqa: E402

dev = torch.device("cuda:0")
steps = int(os.environ.get("CURVE_STEPS", "72"))
W = 6


def windows(c):
    return torch.tensor(c, dtype=torch.float64).view(-1, W).mean(1)


for lr in [float(v) for v in os.environ.get("CURVE_LRS", "0.0005,0.001,0.002").split(",")]:
    curves = {m: windows(T._loss_curve(dev, m, lr, steps)) for m in ("fp32", "mixed", "bf16", "bf16x3")}
    again = windows(T._loss_curve(dev, "fp32", lr, steps))      # run-to-run (atomics order)
    f = curves["fp32"]
    print("lr %g: fp32 pass means %s" % (lr, [round(v, 1) for v in f.tolist()]))
    print("   fp32 again  : %s" % ["%.3f" % v for v in (again / f - 1).tolist()])
    for m in ("mixed", "bf16x3", "bf16"):
        print("   %-12s: %s" % (m, ["%.3f" % v for v in (curves[m] / f - 1).tolist()]))
