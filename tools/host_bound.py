"""Is the eager train step host-bound?  Times K steps of bench.py's train-step workload twice: until the host has
enqueued them (no sync) and until the device has finished; then a cProfile of 5 steps (top by own time).
    python tools/host_bound.py [--profile]"""
import cProfile
import os
import pstats
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
import bench  # noqa: E402

args = SimpleNamespace(batch=32, frames=1000, labels=50, vocab=5000)
R_ = bench.Ranks(False)
step, _ = bench.train_step_workload(args, R_)
try:
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    K = 10
    t0, c0 = time.perf_counter(), time.thread_time()
    for _ in range(K):
        step()
    t1, c1 = time.perf_counter(), time.thread_time()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # wall time until the last step is enqueued includes waiting for the loader's back-pressure (its pinned ring is
    # recycled at the device's pace); the CPU time of this thread is what the step costs the host
    print("launch mode: %s" % ("eager" if os.environ.get("PIKA_TRAIN_GRAPH", "1") == "0" else "forward + backward graph replays behind Net.forward (pika_amd.train_graph)"))
    print("host enqueue (wall) %.2f ms/step, host CPU time of the training thread %.2f ms/step, device done %.2f ms/step"
          % ((t1 - t0) / K * 1e3, (c1 - c0) / K * 1e3, (t2 - t0) / K * 1e3), flush=True)
    # the same with an EMPTY queue in front of every step (the loop above runs ahead of the device until the hardware queue
    # is full, and the runtime spins on a full queue: that wait is CPU time but not work)
    cpu = wall = 0.0
    for _ in range(K):
        torch.cuda.synchronize()
        t0, c0 = time.perf_counter(), time.thread_time()
        step()
        cpu += time.thread_time() - c0
        wall += time.perf_counter() - t0
    torch.cuda.synchronize()
    print("one step at a time, queue empty: host enqueue (wall) %.2f ms/step, host CPU of the training thread %.2f ms/step"
          % (wall / K * 1e3, cpu / K * 1e3), flush=True)
    if "--profile" in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            step()
        pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(45)
finally:
    step.close()
