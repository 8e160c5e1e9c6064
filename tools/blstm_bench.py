"""LAS encoder (2-layer BLSTM 1024 over 64 utterances x 250 positions): persistent-kernel path vs the library nn.LSTM.
    python tools/blstm_bench.py"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pika_amd.model import las  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
S, B = 250, 64
enc = las.LASRNNEncoder("LSTM", True, 2, 1024, 0.0, 1024).to(dev).eval()
x = torch.randn(S, B, 1024, device=dev)
lens = torch.full((B,), S, dtype=torch.int64)


def timeit(n=10):
    with torch.no_grad():
        enc(x, lens); enc(x, lens)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            enc(x, lens)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for mode in ("1", "0"):
    os.environ["PIKA_LAS_BLSTM"] = mode
    print("PIKA_LAS_BLSTM=%s: %.3f ms per encoder pass" % (mode, timeit()), flush=True)
