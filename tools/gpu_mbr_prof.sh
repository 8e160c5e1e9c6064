#!/bin/bash
# where an MBR step (configs[3], B = 8, beam 4) spends its time: cProfile of the host side + kernel-trace summary
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python - > gpurun_out/mbr_host.txt 2>&1 <<'PY'
import cProfile, pstats, sys, os, time
from types import SimpleNamespace
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pika_amd", "dropin"))
import torch, bench
args = SimpleNamespace(batch=8, frames=1000, labels=50, vocab=5000, beam=4)
step, info = bench.mbr_workload(args, torch.device("cuda:0"), 0)
for _ in range(2): step()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter()-t0)/3*1e3, info)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
PY
head -75 gpurun_out/mbr_host.txt | cut -c1-170
