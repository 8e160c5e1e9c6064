"""The script's training loop on a corpus-like stream of batches (the recipes' own operating point: 8 utterances per GPU, V = 6268,
LSTM prediction network; utterance lengths 4-16 s, batches of similar length in random order, as utils/shuffle_by_length.py
makes them): ms per step with the graphed step on (what pika_amd.launch gives a script) and off, and what the graphs did.
    GPU box: python tools/real_corpus_sim.py [steps] [max_graphs]"""
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
import pika_amd  # noqa: E402,F401
import torch  # noqa: E402
from pika_amd import gemm as G, optim as fused_optim, train_graph  # noqa: E402
from pika_amd.model.transducer import Net  # noqa: E402
from warp_rnnt import RNNTLoss  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
max_graphs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
V, B = 6268, 8
opt = SimpleNamespace(rnn_size=1024, local_rank=0, decoder_type="rnn", brnn=False, encoder_type="tdnn", dropout=0.2,
                      enc_layers=9, dec_layers=2, embd_dim=100, padding_idx=V)
g = torch.Generator().manual_seed(3)
# batches: a length bucket per batch (frames), labels ~ 4.5 per second with a little spread; order shuffled
lengths = torch.randint(400, 1601, (steps,), generator=g).tolist()
batches = []
for T in lengths:
    U = max(3, int(T / 100 * 4.5) + int(torch.randint(-2, 3, (1,), generator=g)))
    x = torch.randn(B, T, 240, generator=g).to(dev)
    y = torch.randint(1, V, (B, U), generator=g).to(dev)
    ali = torch.randint(max(1, U - 4), U + 1, (B,), generator=g).int()
    ali[0] = U
    for b in range(B):
        y[b, int(ali[b]):] = V
    x_len = torch.full((B,), (T - 42 + 3) // 4, dtype=torch.int32, device=dev)
    batches.append((x, y, x_len, ali.to(dev)))


def run(auto):
    torch.manual_seed(0)
    model = Net(opt, 240, V).to(dev).train()
    loss_fn = RNNTLoss(blank=0, reduction="sum").apply
    optimizer = torch.optim.SGD(model.parameters(), 1e-4, momentum=0.9, nesterov=True)
    train_graph.AUTO = auto
    train_graph.DEFAULTS["max_graphs"] = max_graphs
    t_steps = []
    for i, (x, y, x_len, ali) in enumerate(batches):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        optimizer.zero_grad()
        out = model.forward(x, y.long(), x_len, True)
        loss = loss_fn(out, y.int(), x_len, ali).sum()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 3.0, norm_type=float("inf"))
        optimizer.step()
        torch.cuda.synchronize()
        t_steps.append(time.perf_counter() - t0)
    st = model.__dict__.get("_step_graphs")
    stats = None if st is None else dict(st.stats, graphs=len(st.entries), broken=st.broken)
    if st is not None:
        train_graph.disable(model)
    half = t_steps[len(t_steps) // 2:]
    return 1e3 * sum(t_steps) / len(t_steps), 1e3 * sum(half) / len(half), stats


G.PRECISION = "mixed"
fused_optim.install()
for auto in (False, True):
    mean, late, stats = run(auto)
    print("graphs %s: %.1f ms per step over %d steps (second half: %.1f ms)   %s" % ("on " if auto else "off", mean, steps, late, stats), flush=True)
