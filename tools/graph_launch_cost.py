"""Host cost of replaying a hipGraph on this ROCm build, by node count and node kind: the host CPU time `graph.replay()`
takes (thread_time) and the wall time until the device is done.    python tools/graph_launch_cost.py"""
import os
import time

import torch

dev = torch.device("cuda:0")
x = torch.zeros(1 << 14, device=dev)
big = torch.zeros(1 << 26, device=dev)


def measure(name, body, reps=20):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
        g.capture_begin()
        body()
        g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    t0, c0 = time.perf_counter(), time.thread_time()
    for _ in range(reps):
        g.replay()
    t1, c1 = time.perf_counter(), time.thread_time()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-44s host wall %7.3f ms  host CPU %7.3f ms  device done %7.3f ms per replay" % (
        name, (t1 - t0) / reps * 1e3, (c1 - c0) / reps * 1e3, (t2 - t0) / reps * 1e3), flush=True)


for n in (100, 600):
    measure("%d tiny kernels" % n, lambda n=n: [x.add_(1.0) for _ in range(n)])
measure("600 tiny kernels + 80 memsets", lambda: [x.add_(1.0) for _ in range(600)] + [x[:256].zero_() for _ in range(80)])
measure("600 kernels of ~70 us each", lambda: [big.add_(1.0) for _ in range(600)])
print("env:", {k: v for k, v in os.environ.items() if "GRAPH" in k})
