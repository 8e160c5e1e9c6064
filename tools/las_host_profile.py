"""cProfile of the host side of the LAS rescoring inside the configs[4] decode leg (bench.decode_workload with --fst --las):
where the ~18 ms of a pass pair that are not device-busy go.    GPU box: python tools/las_host_profile.py"""
import cProfile
import os
import pstats
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
import torch  # noqa: E402
import bench  # noqa: E402

args = SimpleNamespace(batch=64, frames=1000, labels=50, vocab=5000, beam=16, pred_net="transformer", fst=True, las=True,
                       fst_scale=0.3, blank_bias=None, decode_model="speechlike", decode_eager=False)
step, _ = bench.decode_workload(args, torch.device("cuda:0"), 0)
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize()
print("ms per batch %.1f; las calls %s" % ((time.perf_counter() - t0) / 3 * 1e3, step.las_calls[-3:]))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(70)
