"""Debug: per-parameter gradient of one graphed step vs the eager step on the same batch (small model)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
import torch  # noqa: E402
import test_train_step_gpu as T  # noqa: E402
from pika_amd import gemm as G  # noqa: E402
from pika_amd import train_graph  # noqa: E402
from pika_amd.train_graph import GraphedTrainStep  # noqa: E402

V = int(os.environ.get("PROBE_V", "500"))
model, loss_fn, batches, fused_optim = T._small_step_harness("cuda:0", 0.0, V=V)
G.PRECISION = "mixed"
fused_optim.install()
gs = GraphedTrainStep(model, loss_fn, lambda: torch.optim.SGD(model.parameters(), 0.0, momentum=0.9, nesterov=True),
                      clip=0.0, warmup=1)
gs(*batches[0])
for rep in range(2):
    gs(*batches[1])
    print("grad None after the graphed step:", [n for n, p in model.named_parameters() if p.grad is None])
    got = {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}
    print("stats", gs.state.stats)
    train_graph.disable(model)
    model.zero_grad(set_to_none=True)
    b = batches[1]
    loss_fn(model(b[0], b[1].long(), b[2], True), b[1].int(), b[2], b[3]).sum().backward()
    rows = []
    for n, p in model.named_parameters():
        d = (got[n] - p.grad).abs().max().item()
        rows.append((d / (p.grad.abs().max().item() + 1e-30), n, p.grad.abs().max().item(), got[n].abs().max().item()))
    rows.sort(reverse=True)
    for r in rows[:6]:
        print("  %.3e %-50s eager max %.3e graph max %.3e" % r)
    gs.state = train_graph.enable(model, warmup=0)
