"""Full-architecture TRAIN-mode parity (VERDICT r2 missing #2): the model of BASELINE.json configs[1] (1024 wide, 9 TDNN
+ 3 transformer layers, conv-transformer prediction net, V = 5000), BatchNorm on batch statistics, against the golden
recorded from the REFERENCE model (tests/golden/make_model_full_golden.py: trainer/model/transducer.py:73-112 on CPU
fp32, RNN-T costs of its log-probs by the fp64 oracle).  north_star: encoder activations and loss within 1e-3 rel.

Every arithmetic mode of the GPU path is run against the same golden; the tolerances are per mode and written here:

  mode     encoder act.      costs             gradients (||dg|| / ||g|| per parameter, on the recorded samples)
  fp32     1e-4              1e-5              2e-2 encoder (ReLU-mask flips: two CPU fp32 runs differ by 1.5e-2, see the
                                               CPU test below), 1e-3 prediction net + joint
  bf16x3   2e-4 (5.5e-5)     5e-4 (2.1e-4)     4e-2 (1.8e-2 worst, 9.8e-3 median) / 3e-2 (6.9e-3)   (joint lattice products on
                                               one bf16 term, as benchmarked)
  mixed    2e-4 (4.9e-5)     5e-4 (2.2e-4)     4e-2 (1.9e-2 worst, 1.3e-2 median) / 0.2 (0.11; 4.1e-2 without the key
                                               projections)   <- the train-step default
  bf16     5e-2 (2.6e-2)     5e-3 (1.5e-3)     0.6 (0.32 worst, 0.26 median) / 0.2 (0.10; 6.8e-2): reported as "no parity"
(measured on the MI355X in brackets).  With a bf16 backward the worst parameter outside the encoder is the key projection of
the prediction network's second attention layer: its gradient is the small remainder of a nearly shift-invariant softmax.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import model_full_common as F  # noqa: E402
from mbr_hooks import compact  # noqa: E402
from oracle.pika_ref import seeded_state_dict  # noqa: E402  (deterministic weights only)

GOLD = os.path.join(HERE, "golden", "model_full_train.npz")


def rel_max(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.abs(a - b).max() / np.abs(b).max())


def grad_report(got, z):
    """Per parameter: relative L2-norm difference estimated on the recorded samples, and |norm - norm_ref| / norm_ref."""
    rows = []
    gmax = max(float(z["m%03d" % i][0]) for i in range(int(z["n"])))
    for i in range(int(z["n"])):
        k = "%03d" % i
        s, sg = z["s" + k].astype(np.float64), got["s" + k].astype(np.float64)
        n_ref = float(z["m" + k][0])
        if n_ref < 1e-5 * gmax:          # gradients that are zero up to rounding (biases in front of a BatchNorm, key biases)
            continue
        ns = np.linalg.norm(s)
        rel = float(np.linalg.norm(sg - s) / ns) if ns > 0 else 0.0
        rows.append((str(z["names"][i]), rel, abs(float(got["m" + k][0]) - n_ref) / n_ref))
    return rows


def run(device, loss, decoder_type="transformer"):
    from pika_amd.model import transducer
    net = F.build(transducer, seeded_state_dict, decoder_type).to(device)
    x, y, x_len, y_len = [t.to(device) for t in F.inputs()]
    seen = {}
    net.encoder.register_forward_hook(lambda m, i, o: seen.__setitem__("enc", o.detach()))
    net.decoder.register_forward_hook(lambda m, i, o: seen.__setitem__("pred", (o[0] if isinstance(o, tuple) else o).detach()))
    lp = net(x, y, x_len, True)
    costs = loss(lp, y.int(), x_len, y_len)
    costs.sum().backward()
    lp = lp.detach()
    if hasattr(lp, "dense"):
        lp = lp.dense()
    grads = {"g%03d" % i: p.grad.detach().float().cpu().numpy() for i, (n, p) in enumerate(net.named_parameters())}
    got = compact(grads)
    got["n"] = np.array(len(grads))
    return net, seen, lp, costs.detach().double().cpu().numpy(), got


def summarize(tag, net, seen, lp, costs, got, z):
    e_enc = rel_max(F.enc_slice(seen["enc"]), z["enc"])
    e_pred = rel_max(seen["pred"][:, :, ::17], z["pred"])
    e_lp = rel_max(F.lp_slice(lp), z["lp"])
    e_cost = float(np.abs(costs - z["costs"]).max() / np.abs(z["costs"]).max())
    rows = grad_report(got, z)
    enc = [r for r in rows if r[0].startswith("encoder.")]
    rest = [r for r in rows if not r[0].startswith("encoder.")]
    w_enc = max(enc, key=lambda r: r[1])
    w_rest = max(rest, key=lambda r: r[1])
    w_rest2 = max((r for r in rest if "linear_keys" not in r[0]), key=lambda r: r[1])      # (an LSTM net has none)
    med_enc = float(np.median([r[1] for r in enc]))
    print("\n[%s] encoder act %.2e  pred-net act %.2e  log-probs %.2e  costs %.2e | gradients: encoder median %.2e worst "
          "%.2e (%s); prediction net + joint worst %.2e (%s), without key projections %.2e (%s)" % (
              tag, e_enc, e_pred, e_lp, e_cost, med_enc, w_enc[1], w_enc[0], w_rest[1], w_rest[0], w_rest2[1], w_rest2[0]))
    e_bn = max(rel_max(net.state_dict()[k[4:]], z[k]) for k in z.files if k.startswith("buf:"))
    return dict(enc=e_enc, pred=e_pred, lp=e_lp, cost=e_cost, g_enc=w_enc[1], g_enc_med=med_enc, g_rest=w_rest[1],
                g_rest_nokeys=w_rest2[1], bn=e_bn)


def test_cpu_module_tree_matches_reference_full_golden():
    """Host plumbing: our module tree on stock torch CPU ops + the fp64 oracle loss is the reference's computation."""
    from oracle import rnnt as O

    class _Loss(torch.autograd.Function):
        @staticmethod
        def forward(ctx, lp, y, tl, ul):
            c, g = O.rnnt_loss(lp.detach().numpy(), y.numpy(), tl.numpy(), ul.numpy())
            ctx.save_for_backward(torch.from_numpy(g.astype(np.float32)))
            return torch.from_numpy(c.astype(np.float32))

        @staticmethod
        def backward(ctx, go):
            return ctx.saved_tensors[0] * go.view(-1, 1, 1, 1), None, None, None
    torch.set_num_threads(8)
    z = np.load(GOLD)
    r = summarize("cpu", *run("cpu", _Loss.apply), z)
    assert r["enc"] < 1e-4 and r["pred"] < 1e-4 and r["lp"] < 1e-4 and r["cost"] < 1e-5 and r["bn"] < 1e-4, r
    assert r["g_rest"] < 1e-3 and r["g_enc"] < 2e-2, r


GOLD_RNN = os.path.join(HERE, "golden", "model_full_train_rnn.npz")


def test_cpu_module_tree_matches_reference_full_golden_lstm_prediction_net():
    """The recipes' configuration (dec_type=rnn, egs/train_transducer_bmuf_otfaug.sh:32): our module tree with the 2-layer
    LSTM prediction network on stock torch CPU ops is the reference's computation (golden: make_model_full_golden.py rnn)."""
    from oracle import rnnt as O

    class _Loss(torch.autograd.Function):
        @staticmethod
        def forward(ctx, lp, y, tl, ul):
            c, g = O.rnnt_loss(lp.detach().numpy(), y.numpy(), tl.numpy(), ul.numpy())
            ctx.save_for_backward(torch.from_numpy(g.astype(np.float32)))
            return torch.from_numpy(c.astype(np.float32))

        @staticmethod
        def backward(ctx, go):
            return ctx.saved_tensors[0] * go.view(-1, 1, 1, 1), None, None, None
    torch.set_num_threads(8)
    z = np.load(GOLD_RNN)
    r = summarize("cpu, LSTM prediction net", *run("cpu", _Loss.apply, "rnn"), z)
    assert r["enc"] < 1e-4 and r["pred"] < 1e-4 and r["lp"] < 1e-4 and r["cost"] < 1e-5 and r["bn"] < 1e-4, r
    assert r["g_rest"] < 1e-3 and r["g_enc"] < 2e-2, r


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["mixed", "fp32"])
def test_gpu_lstm_prediction_net_against_reference_full_golden(hip_device, mode):
    """The configuration every shipped recipe trains -- TDNN-Transformer encoder + 2-layer LSTM prediction network
    (trainer/model/transducer.py:55-61,93-96) -- at full width in TRAIN mode against the reference's golden: the benchmarked
    "mixed" arithmetic and the exact mode, the tolerances of the conv-transformer rows above (the LSTM itself runs in fp32:
    MIOpen's recurrence; encoder, joint and loss are the HIP path)."""
    from pika_amd import gemm as G
    from pika_amd.rnnt import RNNTLoss
    z = np.load(GOLD_RNN)
    old = G.PRECISION
    G.PRECISION = mode
    try:
        r = summarize(mode + ", LSTM prediction net", *run(hip_device, RNNTLoss(blank=0).apply, "rnn"), z)
    finally:
        G.PRECISION = old
    t_enc, t_cost, t_genc, t_grest, t_grest2 = TOL[mode]
    assert r["enc"] < t_enc and r["pred"] < max(t_enc, 1e-3) and r["cost"] < t_cost, (mode, r)
    assert r["g_enc"] < t_genc and r["g_rest"] < min(t_grest, 8e-2) and r["g_rest_nokeys"] < t_grest2, (mode, r)
    assert r["bn"] < max(t_enc, 1e-3), (mode, r)


TOL = {  # mode: (encoder act, costs, encoder gradients (worst), prediction net + joint gradients: worst, worst without
    #          the key projections)
    "fp32": (1e-4, 1e-5, 2e-2, 1e-3, 1e-3),
    "bf16x3": (2e-4, 5e-4, 4e-2, 3e-2, 3e-2),
    "mixed": (2e-4, 5e-4, 4e-2, 3e-2, 3e-2),      # (the prediction network is an island of the two-term mode: ops.precision_island)
    "bf16": (5e-2, 5e-3, 0.6, 0.2, 0.12),
}


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["mixed", "bf16x3", "fp32", "bf16"])
def test_gpu_modes_against_reference_full_golden(hip_device, mode):
    from pika_amd import gemm as G
    from pika_amd.rnnt import RNNTLoss
    z = np.load(GOLD)
    old = G.PRECISION
    G.PRECISION = mode
    try:
        r = summarize(mode, *run(hip_device, RNNTLoss(blank=0).apply), z)
    finally:
        G.PRECISION = old
    t_enc, t_cost, t_genc, t_grest, t_grest2 = TOL[mode]
    assert r["enc"] < t_enc and r["pred"] < max(t_enc, 1e-3) and r["cost"] < t_cost, (mode, r)
    assert r["g_enc"] < t_genc and r["g_rest"] < t_grest and r["g_rest_nokeys"] < t_grest2, (mode, r)
    assert r["bn"] < max(t_enc, 1e-3), (mode, r)


GOLD_LONG = os.path.join(HERE, "golden", "model_full_train_long.npz")


def test_long_golden_is_the_benchmarked_length():
    """The fixture itself: B = 4 utterances of T_in = 1000 frames -> T' = 240 lattice frames, U = 50 (configs[1]'s length)."""
    z = np.load(GOLD_LONG)
    assert z["enc"].shape[0] == F.LONG.B and z["costs"].shape == (F.LONG.B,)
    assert z["lp"].shape == (F.LONG.B, len(range(0, 240, 13)), len(range(0, 51, 7)), len(range(0, F.V, 61)))
    assert np.isfinite(z["costs"]).all() and (z["costs"] > 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["graph", "eager"])
def test_gpu_mixed_at_the_benchmarked_length_against_reference_golden(hip_device, path):
    """configs[1] at its OWN length: the reference model (trainer/model/transducer.py:74-112,
    rnnt_tdnn_transformer.py:73-89) at T_in = 1000 -- attention over 994 / 976 frames, lattice products over B * 240 * 51
    rows, the tile shapes bench.py runs -- against the benchmarked "mixed" arithmetic, once as the eager launch sequence and
    once through the hipGraph pair behind Net.forward (pass 3 and 4 are REPLAYS; the graph's costs / log-probs / gradients
    are compared, the activations come from the eager pass of the same arithmetic since a replay runs no module hooks).
    Tolerances: TOL["mixed"] -- the ones of the T_in = 420 golden."""
    from pika_amd import gemm as G
    from pika_amd import train_graph
    from pika_amd.model import transducer
    from pika_amd.rnnt import RNNTLoss
    z = np.load(GOLD_LONG)
    old, old_auto = G.PRECISION, train_graph.AUTO
    G.PRECISION = "mixed"
    loss = RNNTLoss(blank=0).apply
    try:
        net = F.build(transducer, seeded_state_dict).to(hip_device)
        x, y, x_len, y_len = [t.to(hip_device) for t in F.inputs(F.LONG)]
        seen = {}
        hooks = [net.encoder.register_forward_hook(lambda m, i, o: seen.__setitem__("enc", o.detach().clone())),
                 net.decoder.register_forward_hook(
                     lambda m, i, o: seen.__setitem__("pred", (o[0] if isinstance(o, tuple) else o).detach().clone()))]
        bn0 = None
        n_pass = 4 if path == "graph" else 1
        if path == "graph":
            st = train_graph.enable(net, warmup=1, min_seen=1)
        for k in range(n_pass):
            net.zero_grad(set_to_none=True)
            lp = net.forward(x, y, x_len, True)
            costs = loss(lp, y.int(), x_len, y_len)
            costs.sum().backward()
            if k == 0:          # (no hooks inside the capture) the golden's BatchNorm statistics are those after ONE step
                for h in hooks:
                    h.remove()
                bn0 = {k2: net.state_dict()[k2[4:]].detach().clone() for k2 in z.files if k2.startswith("buf:")}
        if path == "graph":
            assert st.broken is None, st.broken
            assert st.stats["captures"] == 1 and st.stats["replays"] >= 2, st.stats
            assert next(iter(st.entries.values())).kind == "compact"
        lp = lp.detach()
        if hasattr(lp, "dense"):
            lp = lp.dense()
        grads = {"g%03d" % i: p.grad.detach().float().cpu().numpy() for i, (n, p) in enumerate(net.named_parameters())}
        got = compact(grads)
        got["n"] = np.array(len(grads))
        costs = costs.detach().double().cpu().numpy()
        if path == "graph":
            train_graph.disable(net)
    finally:
        G.PRECISION = old
        train_graph.AUTO = old_auto
    e_enc = rel_max(F.enc_slice(seen["enc"]), z["enc"])
    e_pred = rel_max(seen["pred"][:, :, ::17], z["pred"])
    e_lp = rel_max(F.lp_slice_long(lp), z["lp"])
    e_cost = float(np.abs(costs - z["costs"]).max() / np.abs(z["costs"]).max())
    rows = grad_report(got, z)
    enc = [r for r in rows if r[0].startswith("encoder.")]
    rest = [r for r in rows if not r[0].startswith("encoder.")]
    w_enc, w_rest = max(enc, key=lambda r: r[1]), max(rest, key=lambda r: r[1])
    e_bn = max(rel_max(bn0[k2], z[k2]) for k2 in bn0)
    print("\n[mixed, T_in = 1000, %s] encoder act %.2e  pred-net act %.2e  log-probs %.2e  costs %.2e | gradients: encoder "
          "median %.2e worst %.2e (%s); prediction net + joint worst %.2e (%s); BatchNorm statistics %.2e" % (
              path, e_enc, e_pred, e_lp, e_cost, float(np.median([r[1] for r in enc])), w_enc[1], w_enc[0], w_rest[1],
              w_rest[0], e_bn))
    t_enc, t_cost, t_genc, t_grest, _ = TOL["mixed"]
    # log-prob SAMPLES: the lattice product runs on one bf16 term per operand in "mixed" (section 5.1b of DESIGN.md: h is a bf16
    # tensor anyway), i.e. 2^-9 per operand on logits of |x| <= 30 -- measured 1.3e-3 of the largest log-prob; the COSTS
    # (sums over ~290 lattice cells) are what north_star bounds at 1e-3 and sit at 9e-5
    assert e_enc < t_enc and e_pred < max(t_enc, 1e-3) and e_cost < t_cost and e_lp < 3e-3, (e_enc, e_pred, e_cost, e_lp)
    assert w_enc[1] < t_genc and w_rest[1] < t_grest, (w_enc, w_rest)
    assert e_bn < max(t_enc, 1e-3), e_bn
