"""Host clock of the stages of one LAS rescoring call (pika_amd.model.las._run_stages) inside the configs[4] decode leg: when
the host reaches every stage boundary of the forward and the backward pass, without device waits other than the passes' own.
    GPU box: python tools/las_stage_clock.py"""
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
import torch  # noqa: E402
import bench  # noqa: E402
from pika_amd.model import las  # noqa: E402

LOG = []


def clocked(gens):
    """The product's _run_stages on generators that log when each of their stages ends."""
    t0 = time.perf_counter()
    names = ["A: plan (first pass: encoder queued first)", "B: encoder (later passes) + uploads + loop prepared", "C: tail queued",
             "D: waited + lists"]

    def wrap(k, g):
        for name in names:
            v = next(g)
            LOG.append(("%s%d" % (name[0], k) + name[1:], (time.perf_counter() - t0) * 1e3))
            yield v
    orig_drive = las.drive_token_loops

    def drive(loops, between=None):
        orig_drive(loops, between=between)
        LOG.append(("   token loop queued (capture + replays)", (time.perf_counter() - t0) * 1e3))
    las.drive_token_loops = drive
    try:
        return ORIG([wrap(k, g) for k, g in enumerate(gens)])
    finally:
        las.drive_token_loops = orig_drive


ORIG = las._run_stages
run_stages = clocked
las._run_stages = run_stages
args = SimpleNamespace(batch=64, frames=1000, labels=50, vocab=5000, beam=16, pred_net="transformer", fst=True, las=True,
                       fst_scale=0.3, blank_bias=None, decode_model="speechlike", decode_eager=False)
step, _ = bench.decode_workload(args, torch.device("cuda:0"), 0)
for _ in range(3):
    step()
torch.cuda.synchronize()
del LOG[:]
step()
torch.cuda.synchronize()
print("las call %.1f ms" % (step.las_calls[-1] * 1e3))
prev = 0.0
for name, t in sorted(LOG, key=lambda e: e[1]):
    print("  %7.2f ms (+%5.2f)  %s" % (t, t - prev, name))
    prev = t
