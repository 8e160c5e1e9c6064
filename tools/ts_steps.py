"""Per-step wall time and caching-allocator counters of the train-step workload (bench.py) on the GPU box."""
import os
import sys
import time
from types import SimpleNamespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402

args = SimpleNamespace(batch=32, frames=1000, labels=50, vocab=5000)
dev = torch.device("cuda:0")
bench._dropin_paths() if hasattr(bench, "_dropin_paths") else None
step, _ = bench.train_step_workload(args, dev, 0, 1)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    st = torch.cuda.memory_stats()
    print("step %d: %7.1f ms  reserved %.1f GB  allocated peak %.1f GB  alloc_retries %d  segments %d  device mallocs %d frees %d" % (
        i, dt, st["reserved_bytes.all.current"] / 2**30, st["allocated_bytes.all.peak"] / 2**30, st["num_alloc_retries"],
        st["segment.all.current"], st["segment.all.allocated"], st["segment.all.freed"]), flush=True)
