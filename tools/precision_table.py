#!/usr/bin/env python
"""Arithmetic mode -> measured error (GPU).  Two panels, printed as markdown:

A. vs the REFERENCE golden (tests/golden/model_tiny_transformer.npz, recorded from the reference model code on
   PyTorch-CPU fp32): encoder activations, joint log-probs, and the 12+3 recorded parameter gradients.
B. full config-2 architecture (1024 wide, 9 TDNN + 3 transformer layers, V = 5000), B = 8: bf16 mode against the
   fp32-exact mode (the one panel A pins): encoder output, per-utterance RNN-T cost, every parameter gradient.

Error measure: max |a - b| / max |b| per tensor (what tests/test_model.py::close asserts at 1e-3 for north_star).
    python tools/precision_table.py > profiles/r2_precision_table.md"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
import model_common as C  # noqa: E402
from test_model import ours  # noqa: E402
from pika_amd import gemm as G  # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


print("## A. tiny model vs the reference golden (PyTorch-CPU fp32 of the reference's own model code)\n")
print("| mode | encoder act. (eval) | joint log-probs (eval) | joint log-probs (train) | worst of the recorded gradients |")
print("|---|---|---|---|---|")
z = np.load(os.path.join(ROOT, "tests", "golden", "model_tiny_transformer.npz"))
for mode in ("fp32", "bf16x3-full", "bf16x3", "bf16"):
    G.PRECISION = mode.split("-")[0]
    G.X3_JOINT_BF16 = mode != "bf16x3-full"
    net = ours("transformer", dev)
    x, y, y_len, w = [t.to(dev) for t in C.inputs()]
    net.eval()
    with torch.no_grad():
        e_enc = rel(net.encoder(x), z["enc_eval"])
        e_joint = rel(net(x, y, None, True), z["joint_eval"])
    net.train()
    lp = net(x, y, None, True)
    e_train = rel(lp, z["joint_train"])
    (lp * w).sum().backward()
    params = dict(net.named_parameters())
    # gradients that are zero up to rounding in the reference (biases in front of a BatchNorm / LayerNorm shift) are
    # compared on the scale of the largest recorded gradient, as tests/test_model.py does with its atol
    gscale = max(float(np.abs(z["grad:" + str(k)]).max()) for k in z["grad_keys"])
    worst = max((float(np.abs(params[str(k)].grad.detach().cpu().numpy() - z["grad:" + str(k)]).max()
                       / max(float(np.abs(z["grad:" + str(k)]).max()), 1e-3 * gscale)), str(k)) for k in z["grad_keys"])
    print("| %s | %.2e | %.2e | %.2e | %.2e (%s) |" % ({"fp32": "fp32-exact (3-term split, 6 MFMAs)", "bf16": "bf16 operands",
                                                       "bf16x3-full": "bf16x3, every product (2 terms per operand, 3 products in one bf16 GEMM)",
                                                       "bf16x3": "bf16x3 as benchmarked (the joint's lattice products on bf16 operands)"}[mode],
                                                      e_enc, e_joint, e_train, worst[0], worst[1]))

print("\n## B. full config-2 architecture, B = 8, T_in = 420: bf16 mode vs fp32-exact mode\n")
from pika_amd.model.transducer import Net  # noqa: E402
from pika_amd.rnnt import RNNTLoss  # noqa: E402
B, T, U, V = 8, 420, 12, 5000
opt = SimpleNamespace(rnn_size=1024, local_rank=0, decoder_type="transformer", brnn=False, encoder_type="tdnn",
                      dropout=0.0, enc_layers=4, dec_layers=2, embd_dim=100, padding_idx=V)
torch.manual_seed(5)
model = Net(opt, 240, V).to(dev).train()
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


def run(mode, x=None):
    x = data if x is None else x
    model.load_state_dict(bn_state, strict=False)
    model.zero_grad(set_to_none=True)
    G.PRECISION = mode.split("-")[0]
    G.X3_JOINT_BF16 = mode != "bf16x3-full"
    enc = model.encode(x, None)
    out = model(x, labels, len_b, True)
    costs = RNNTLoss(blank=0).apply(out, labels.int(), len_b, ali)
    costs.sum().backward()
    return enc.detach(), costs.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


e32, c32, g32 = run("fp32")
cols = {}
# control: the EXACT mode on inputs perturbed by 2e-5 relative -- what a forward difference of the bf16x3 size does
# to the gradients of this ReLU / BatchNorm network irrespective of the arithmetic
noise = torch.randn(data.shape, generator=torch.Generator().manual_seed(99)).to(dev)
for mode in ("bf16x3-full", "bf16x3", "bf16", "control"):
    e16, c16, g16 = run("fp32", data * (1 + 2e-5 * noise)) if mode == "control" else run(mode)
    grel = {n: float(((g16[n] - g32[n]).double().norm() / g32[n].double().norm().clamp_min(1e-30))) for n in g32
            if float(g32[n].double().norm()) > 1e-4 * max(1.0, g32[n].numel() ** 0.5)}
    enc_g = {n: v for n, v in grel.items() if n.startswith("encoder.")}
    oth_g = {n: v for n, v in grel.items() if not n.startswith("encoder.")}
    cols[mode] = ["%.2e" % rel(e16, e32), "%.2e" % float(((c16 - c32).abs() / c32.abs()).max()),
                  "%.2e (%s)" % max((v, n) for n, v in oth_g.items()), "%.2e (%s)" % max((v, n) for n, v in enc_g.items()),
                  "%.2e" % float(np.median(list(enc_g.values())))]
print("| quantity | bf16x3 (every product) vs fp32-exact | bf16x3 as benchmarked vs fp32-exact | bf16 vs fp32-exact | fp32-exact on inputs x (1 + 2e-5 N(0,1)) vs fp32-exact (control) |")
print("|---|---|---|---|---|")
for i, q in enumerate(["encoder output (B,T',1024), max abs err / max abs", "RNN-T cost per utterance, max rel",
                       "prediction-net / joint parameter gradients, worst ||dg||/||g||",
                       "encoder parameter gradients, worst ||dg||/||g||", "encoder parameter gradients, median ||dg||/||g||"]):
    print("| %s | %s | %s | %s | %s |" % (q, cols["bf16x3-full"][i], cols["bf16x3"][i], cols["bf16"][i], cols["control"][i]))
