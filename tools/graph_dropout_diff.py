"""Debug / evidence: with dropout ON, the gradients of a replayed step equal those of the eager step that draws the same
masks (same host seeds via torch.manual_seed, dropout salt frozen)."""
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

V = int(os.environ.get("PROBE_V", "512"))
if os.environ.get("PROBE_FULL"):
    # the architecture the training script builds (1024 wide, 9 TDNN + 3 transformer layers, fused attention with dropout)
    from types import SimpleNamespace
    from pika_amd.model.transducer import Net
    from pika_amd.rnnt import RNNTLoss
    V = 5000
    opt = SimpleNamespace(rnn_size=1024, local_rank=0, decoder_type="transformer", brnn=False, encoder_type="tdnn",
                          dropout=0.2, enc_layers=4, dec_layers=2, embd_dim=100, padding_idx=V)
    torch.manual_seed(5)
    model = Net(opt, 240, V).to("cuda:0").train()
    g = torch.Generator().manual_seed(6)
    b = T._batch("cuda:0", g, 8, 389, 8, V, pad_from=5)
    loss_fn = RNNTLoss(blank=0).apply
else:
    model, loss_fn, batches, fused_optim = T._small_step_harness("cuda:0", 0.2, V=V)
    b = batches[1]
G.PRECISION = os.environ.get("PROBE_MODE", "mixed")


def eager():
    model.zero_grad(set_to_none=True)
    torch.manual_seed(1234)
    loss = loss_fn(model._forward_eager(b[0], b[1].long(), b[2], True), b[1].int(), b[2], b[3]).sum()
    loss.backward()
    return float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()}


st = train_graph.enable(model, warmup=0, min_seen=1)
st.freeze_salt = True
l0, g0 = eager()
l0b, g0b = eager()
print("eager twice: loss %.6f %.6f" % (l0, l0b))
for rep in range(3):
    model.zero_grad(set_to_none=True)
    torch.manual_seed(1234)          # the capture draws the host seeds the eager run drew
    loss = loss_fn(model(b[0], b[1].long(), b[2], True), b[1].int(), b[2], b[3]).sum()
    loss.backward()
    rows = []
    for n, p in model.named_parameters():
        d = (p.grad - g0[n]).abs().max().item()
        rows.append((d / (g0[n].abs().max().item() + 1e-30), n))
    rows.sort(reverse=True)
    print("replay %d: loss %.6f (eager %.6f) stats %s; worst relative gradient differences: %s" % (
        rep, float(loss), l0, st.stats, ["%.2e %s" % r for r in rows[:4]]))

if os.environ.get("PROBE_STEPS"):
    # several optimisation steps on DIFFERENT batches, dropout on, same masks on both sides (host seeds re-seeded per step,
    # salt frozen): the loss sequences of the replayed and of the eager loop must agree
    import copy
    from pika_amd import optim as fused_optim2
    fused_optim2.install()
    train_graph.disable(model)
    ref = copy.deepcopy(model)
    g = torch.Generator().manual_seed(77)
    if os.environ.get("PROBE_FULL"):
        bs = [T._batch("cuda:0", g, 8, 387, 8, V, pad_from=5) for _ in range(int(os.environ.get("PROBE_STEPS")))]
        if os.environ.get("PROBE_RAGGED_T"):        # utterances of a batch end at different frames, as in the corpus
            bs = [(a, b_, torch.tensor([87, 78, 71, 87, 78, 71, 87, 78], dtype=torch.int32, device="cuda:0"), d)
                  for a, b_, c, d in bs]
    else:
        bs = [T._batch("cuda:0", g, 4, 300, 8, V, pad_from=5) for _ in range(8)]

    def loop(m):
        o = torch.optim.SGD(m.parameters(), 0.003, momentum=0.9, nesterov=True)
        out = []
        for bb in bs:
            o.zero_grad()
            torch.manual_seed(4321)
            loss = loss_fn(m(bb[0], bb[1].long(), bb[2], True), bb[1].int(), bb[2], bb[3]).sum()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 3.0, norm_type=float("inf"))
            o.step()
            out.append(round(float(loss), 3))
        return out
    rot = int(os.environ.get("PROBE_ROTATE", "0"))
    bs = bs[rot:] + bs[:rot]
    want = loop(ref)
    st = train_graph.enable(model, warmup=int(os.environ.get("PROBE_WARMUP", "1")), min_seen=1)
    st.freeze_salt = True
    got = loop(model)
    print("eager :", want)
    print("graphs:", got, st.stats)
