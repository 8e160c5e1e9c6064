"""The 312-step loop of tests/test_train_step_gpu.py::test_mixed_arithmetic_tracks_fp32_over_300_steps, N times in "fp32" and
N times in "mixed" (and in "bf16"): per pass over the 24 recurring batches, the distance |a / b - 1| between the per-pass
mean losses of every PAIR of runs -- fp32 against fp32 (the band: the runs differ in the order of float atomics only),
mixed against fp32, bf16 against fp32 -- as median and maximum over the pairs.  Where the trajectories have parted ways
(pass 5 on) an arithmetic can only be asked to sit as far from an fp32 run as another fp32 run does.
GPU box: python tools/mixed_band.py [N]"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pika_amd  # noqa: E402,F401
import torch  # noqa: E402

import test_train_step_gpu as T  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
steps, nb = 312, 24


def run(mode):
    return torch.tensor(T._loss_curve(dev, mode, T.CURVE_LR, steps, n_batches=nb), dtype=torch.float64).view(-1, nb).mean(1)


runs = {m: [run(m) for _ in range(N)] for m in ("fp32", "mixed", "bf16")}


def dist(pairs):
    d = torch.stack([(a / b - 1).abs() for a, b in pairs])
    return d.median(0).values, d.max(0).values


def line(tag, med, mx):
    print("%-16s median %s" % (tag, " ".join("%7.1e" % v for v in med.tolist())))
    print("%-16s max    %s" % ("", " ".join("%7.1e" % v for v in mx.tolist())))


print("# %d runs per arithmetic, %d steps = %d passes over %d batches, lr %g; columns = passes" % (N, steps, steps // nb, nb, T.CURVE_LR))
f = torch.stack(runs["fp32"])
print("fp32 per-pass loss, mean over runs: %s" % " ".join("%7.1f" % v for v in f.mean(0).tolist()))
line("fp32  vs fp32", *dist(list(itertools.combinations(runs["fp32"], 2))))
line("mixed vs fp32", *dist([(a, b) for a in runs["mixed"] for b in runs["fp32"]]))
line("mixed vs mixed", *dist(list(itertools.combinations(runs["mixed"], 2))))
line("bf16  vs fp32", *dist([(a, b) for a in runs["bf16"] for b in runs["fp32"]]))
for m in ("fp32", "mixed", "bf16"):
    last = torch.stack(runs[m])[:, -1]
    print("last pass, %-5s: mean %.2f, min %.2f, max %.2f over the %d runs" % (m, last.mean(), last.min(), last.max(), N))
