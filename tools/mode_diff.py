"""Per-parameter gradient difference between the bf16 arithmetic mode and the fp32 parity mode on the config-2
architecture (diagnostic for tests/test_train_step_gpu.py).  PIKA_DIFF_MODES="a,b" picks the pair."""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
from pika_amd import gemm as G  # noqa: E402
from pika_amd.model.transducer import Net  # noqa: E402
from pika_amd.rnnt import RNNTLoss  # noqa: E402

dev = torch.device("cuda:0")
B, T, U, V = 8, 420, 12, 5000
opt = SimpleNamespace(rnn_size=1024, local_rank=0, decoder_type="transformer", brnn=False, encoder_type="tdnn",
                      dropout=0.0, enc_layers=4, dec_layers=2, embd_dim=100, padding_idx=V)
torch.manual_seed(5)
model = Net(opt, 240, V).to(dev)
model.train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
g = torch.Generator().manual_seed(6)
data = torch.randn(B, T, 240, generator=g).to(dev)
labels = torch.randint(1, V, (B, U), generator=g).to(dev)
lens = torch.tensor([T - 7 * i for i in range(B)], dtype=torch.int32, device=dev)
len_b = lens - 42
len_b = len_b // 4 + (len_b % 4 != 0).int()
ali = torch.tensor([U - (i % 3) for i in range(B)], dtype=torch.int32, device=dev)
bn_state = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}


def run(mode):
    model.load_state_dict(bn_state, strict=False)
    model.zero_grad(set_to_none=True)
    os.environ.pop("PIKA_NO_FUSED", None)
    if mode == "bf16-plain":
        os.environ["PIKA_NO_FUSED"] = "1"
        mode = "bf16"
    G.PRECISION = mode
    if os.environ.get("PIKA_DIFF_LOSS") == "enc":   # encoder only, zero-mean random cotangent (no common mode)
        enc = model.encoder(data)
        gw = torch.randn(enc.shape, generator=torch.Generator().manual_seed(9)).to(dev)
        costs = (enc * gw).sum(dim=(1, 2))
    else:
        out = model(data, labels, len_b, True)
        costs = RNNTLoss(blank=0).apply(out, labels.int(), len_b, ali)
    costs.sum().backward()
    return costs.detach().double().cpu(), {n: p.grad.detach().double().cpu() for n, p in model.named_parameters()
                                           if p.grad is not None}


ma, mb = os.environ.get("PIKA_DIFF_MODES", "bf16,fp32").split(",")
ca, ga = run(ma)
cb, gb = run(mb)
print("costs", ma, ca.tolist())
print("costs", mb, cb.tolist())
for n in gb:
    a, b = ga[n], gb[n]
    nb = b.norm().item()
    rel = ((a - b).norm() / max(nb, 1e-30)).item()
    cos = (a.flatten() @ b.flatten() / (a.norm() * b.norm()).clamp_min(1e-30)).item()
    print("%-60s |g|=%.3e rel=%.4f cos=%.5f" % (n, nb, rel, cos))
