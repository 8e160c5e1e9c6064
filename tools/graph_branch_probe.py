"""Do two branches of ONE hipGraph overlap on this runtime?  Branch A: a chain of 12 small dependent products (few CUs each);
branch B: one large product.  Replay time ~ max(A, B) if they overlap, A + B if the runtime serialises them.
    python tools/graph_branch_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pika_amd  # noqa: F401,E402

dev = torch.device("cuda:0")
a = torch.randn(160, 512, device=dev)
w = [torch.randn(512, 512, device=dev) * 0.04 for _ in range(12)]
big_a = torch.randn(8192, 2048, device=dev, dtype=torch.bfloat16)
big_b = torch.randn(2048, 8192, device=dev, dtype=torch.bfloat16)


def chain():
    x = a
    for m in w:
        x = torch.relu(x @ m)
    return x


def big():
    return big_a @ big_b


def timeit(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def capture(parallel, which="both"):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    for _ in range(2):
        chain(), big()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        if which in ("both", "chain") and not parallel:
            chain()
        if which in ("both", "big"):
            if parallel:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    big()
                chain() if which == "both" else None
                cur.wait_stream(side)
            else:
                big()
    return g


for name, g in (("chain only", capture(False, "chain")), ("big only", capture(False, "big")),
                ("serial: chain then big", capture(False)), ("two branches of one graph", capture(True))):
    print("%-28s %8.1f us per replay" % (name, timeit(g.replay)))
