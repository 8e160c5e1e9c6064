"""MBR step: the native formulation (pika_amd/mbr.py: device trajectories, split
joint, sparse risk surrogate) against a literal restatement of the reference's inline code
(train_transducer_mbr_bmuf_otfaug.py:169-235: dense one-hot gradient, concatenated joint input,
Python trajectory loops) on the same model, N-best lists and scores."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import model_common as C  # noqa: E402
import decode_common as D  # noqa: E402
from test_decode import build  # noqa: E402


def reference_mbr(model, x_feats, target, ali_lens, hyps, scores, beam, blk, sm_scale, pad):
    """Lines :124-138 (encoder) and :163-235, restated with the reference's tensor ops."""
    import editdistance
    x = model.encoder(x_feats)
    T = x.size(1)
    bsz = x.size(0)
    bb = bsz * beam
    x = x.unsqueeze(1).expand(-1, beam, -1, -1).contiguous().view(bb, -1, model.hid_dim)
    prob = F.softmax(torch.tensor([[float(s) for s in r] for r in scores]).view(bsz, beam).to(x.device), dim=1)
    dist = torch.zeros(bsz, beam, device=x.device)
    nb, maxl = [], 0
    for i in range(bsz):
        nb.append([[int(e) for e in hyps[i][j] if int(e) != blk] for j in range(beam)])
        maxl = max(maxl, max(len(h) for h in nb[i]))
    U = maxl + 1
    for i in range(bsz):
        ref = target[i][:ali_lens[i]].tolist()
        for j in range(beam):
            dist[i][j] = editdistance.eval(ref, nb[i][j])
            nb[i][j] = nb[i][j] + (maxl - len(nb[i][j])) * [pad]
    avg = (prob * dist).sum(dim=1)
    mbr_loss = avg.sum()
    seq_grad = prob * (dist - avg.unsqueeze(1))
    y = torch.tensor(nb, device=x.device).view(bb, -1)
    y = torch.cat((torch.zeros(bb, 1, dtype=torch.long, device=x.device), y), dim=1)
    y = model.predict(y)
    b_idx, x_idx, y_idx = [], [], []
    mbr_grad = x.new_zeros(bb, T + U, model.output_dim)
    for i in range(bsz):
        for j in range(beam):
            h = [int(e) for e in hyps[i][j]]
            t_idx, u_idx = [0], [0]
            for t in range(1, len(h)):
                t_idx.append(t_idx[t - 1] + int(h[t - 1] == blk))
                u_idx.append(u_idx[t - 1] + int(h[t - 1] != blk))
            t_idx.extend((T + U - len(t_idx)) * [0]); u_idx.extend((T + U - len(u_idx)) * [0])
            x_idx.extend(t_idx); y_idx.extend(u_idx); b_idx.extend([i * beam + j] * (T + U))
            if h:
                mbr_grad[i * beam + j, torch.arange(len(h)), h] = seq_grad[i][j]
    joint = torch.cat((x[b_idx, x_idx, :], y[b_idx, y_idx, :]), dim=-1).view(bb, T + U, -1)
    out = model.fc2(torch.tanh(model.fc1(joint)) * torch.sigmoid(model.fc_gate(joint)))
    out = F.log_softmax(sm_scale * out, dim=-1)
    mbr_grad[:, :, blk] = mbr_grad[:, :, blk] / float(T)
    out.backward(mbr_grad)
    return float(mbr_loss), seq_grad.detach()


def run(device):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from decoder.transducer_decoder import TransducerDecoder
    from decoder.beam_transducer import GlobalScorer
    from pika_amd import mbr
    net = build("transformer", device)
    x, y, y_len, _ = [t.to(device) for t in C.inputs()]
    _, x_len = D.inputs()
    beam, blk, sm = 3, 0, 0.9
    args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=0.0)
    net.eval()
    dec = TransducerDecoder(net, 4, beam, n_best=beam, blk=blk, global_scorer=GlobalScorer(), sm_scale=sm,
                            cuda=(device != "cpu"), args=args)
    ret, _ = dec.decode_batch(x, x_len.to(device), [int(a) + int(b) + 3 for a, b in zip(x_len, y_len)])  # :114
    hyps, scores = ret["predictions"], ret["scores"]
    assert any(0 < sum(1 for e in h if int(e) != blk) for row in hyps for h in row)
    net.train()
    # reference formulation
    net.zero_grad()
    loss_ref, sg_ref = reference_mbr(net, x, y, y_len, hyps, scores, beam, blk, sm, C.V)
    g_ref = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    # native formulation
    net.zero_grad()
    enc = net.encoder(x)
    prob, dist, seq_grad, nonblk = mbr.risk_terms(hyps, scores, y, y_len, blk, enc.device)
    assert torch.allclose(seq_grad, sg_ref, atol=1e-6)
    assert abs(float((prob * dist).sum()) - loss_ref) < 1e-4
    mbr.mbr_backward(net, enc, hyps, seq_grad, nonblk, blk, sm)
    for k, p in net.named_parameters():
        if k in g_ref:
            scale = g_ref[k].abs().max().item()
            err = (p.grad - g_ref[k]).abs().max().item()
            assert err <= 2e-3 * scale + 1e-6, (k, err, scale)
    assert set(g_ref) == {k for k, p in net.named_parameters() if p.grad is not None}


def test_cpu_native_mbr_equals_reference_formulation():
    run("cpu")


@pytest.mark.gpu
def test_gpu_native_mbr_equals_reference_formulation(hip_device):
    from pika_amd import gemm as G
    old, G.PRECISION = G.PRECISION, "fp32"
    try:
        run(hip_device)
    finally:
        G.PRECISION = old


@pytest.mark.gpu
def test_risk_grad_kernel(hip_device):
    from pika_amd.mbr import RiskFn
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(37, 100, generator=g, dtype=torch.float64)
    sym = torch.randint(0, 100, (37,), generator=g)
    val = torch.randn(37, generator=g, dtype=torch.float64)
    val[5] = 0.0
    lr = logits.clone().requires_grad_(True)
    (F.log_softmax(0.8 * lr, -1).gather(1, sym.unsqueeze(1)).squeeze(1) * val).sum().backward()
    ld = logits.float().to(hip_device).requires_grad_(True)
    s = RiskFn.apply(ld * 1.0, sym.int().to(hip_device), val.float().to(hip_device), 0.8)
    s.backward()
    assert (ld.grad.double().cpu() - lr.grad).abs().max() < 1e-5
    assert torch.all(ld.grad[5] == 0)


# ---- pinned against the reference SCRIPT itself (tests/golden/make_mbr_script_golden.py) ---------------------------
GOLD_SCRIPT = os.path.join(HERE, "golden", "mbr_script_grads.npz")
MBR_SCRIPT = "/root/reference/trainer/train_transducer_mbr_bmuf_otfaug.py"


@pytest.mark.skipif(not os.path.exists(MBR_SCRIPT), reason="reference tree not present on this box")
def test_unchanged_mbr_script_on_the_dropins_reproduces_the_reference_gradients(tmp_path):
    """The UNCHANGED train_transducer_mbr_bmuf_otfaug.py, one batch, every parameter gradient before its first
    optimizer step: run on the drop-in packages (CPU tensors) vs the golden recorded from the same script on the
    reference's own trainer.model.* / decoder.* -- same N-best, gradients to 1e-3."""
    import subprocess
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import mbr_hooks as M
    from bmuf_common import free_port
    out = str(tmp_path / "dropin.npz")
    env = dict(os.environ, MASTER_PORT=str(free_port()), OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "mbr_hooks.py"), "dropin", out], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    got, want = np.load(out), np.load(GOLD_SCRIPT)
    assert np.array_equal(got["hyps"], want["hyps"])
    assert np.abs(got["scores"] - want["scores"]).max() < 1e-4
    M.compare(got, want, rel=1e-3)


def _native_step_vs_script_golden(device, search_precision=None, _raw=False, graphed=False):
    """pika_amd.mbr (device trajectories, split joint, sparse risk surrogate / HIP risk-gradient kernel) fed with
    the same seeded model and fixture batch as the golden run of the reference script: same N-best out of the
    drop-in decoder, and RNN-T + risk gradients equal to what the script's inline code produced."""
    import argparse
    sys.path.insert(0, os.path.join(HERE, "golden"))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    import importlib.util
    import mbr_hooks as M
    spec = importlib.util.spec_from_file_location(
        "mbr_fixture_loader", os.path.join(HERE, "golden", "mbr_fixture", "loader", "otf_utt_loader.py"))
    fixture = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fixture)
    from oracle.pika_ref import seeded_state_dict
    from model.transducer import Net
    from decoder.transducer_decoder import TransducerDecoder
    from decoder.beam_transducer import GlobalScorer
    from pika_amd import mbr
    from pika_amd.model import ops
    want = np.load(GOLD_SCRIPT)
    ap = argparse.ArgumentParser()
    for k in ("--encoder_type", "--decoder_type", "--rnn_type"):
        ap.add_argument(k)
    for k in ("--enc_layers", "--dec_layers", "--rnn_size", "--embd_dim", "--padding_idx", "--output_dim"):
        ap.add_argument(k, type=int)
    ap.add_argument("--dropout", type=float)
    opt = ap.parse_args(M.MODEL_ARGS)
    opt.local_rank, opt.brnn = 0, False
    net = Net(opt, 240, M.V)
    net.encoder = type(net.encoder)(240, 0, opt.rnn_size, tdnn_nhid=64, tdnn_layers=6)
    net.pack_seq = False
    sd = seeded_state_dict(net, 1234)
    sd["fc2.bias"][0] += 1.5
    net.load_state_dict(sd)
    net = net.to(device)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
            m.dropout = 0.0
    largs = SimpleNamespace(feats_dim=80, lctx=1, rctx=1, batch_size=3, fixture_seed=31, fixture_batches=1,
                            output_dim=M.V, padding_idx=M.V)
    data, target, lens, ali = next(iter(fixture.dataloader(None, None, None, largs)))
    data, target, ali = data.to(device), target.long().to(device), ali.to(device)
    len_b = lens.to(device) - 24
    len_b = len_b // 4 + (len_b % 4 != 0).int()
    beam, blk, sm, rnnt_scale = 3, 0, 0.9, 0.1
    dargs = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=0.0)
    net.eval()
    dec = TransducerDecoder(net, 3, beam, n_best=beam, blk=blk, global_scorer=GlobalScorer(), sm_scale=sm,
                            cuda=(device != "cpu"), beam_prune=False, args=dargs)
    if search_precision is not None:
        dec.decode_precision = search_precision
    ret, _ = dec.decode_batch(data, len_b, len_b + ali + 3)                         # :112-117
    hyps, scores = ret["predictions"], ret["scores"]
    L = want["hyps"].shape[2]
    got_h = np.full(want["hyps"].shape, -1, np.int64)
    for b, row in enumerate(hyps):
        for j, h in enumerate(row):
            got_h[b, j, :len(h)] = [int(e) for e in h]
    score_err = np.abs(np.array([[float(v) for v in r] for r in scores]) - want["scores"]).max()
    if search_precision == "bf16":
        # one bf16 term per operand carries no parity claim (8-bit mantissas: near-ties between deep entries trade places with
        # any change of the product order, e.g. linear_out folded into the joint's prediction halves): the top-1 hypotheses
        # are asserted, the rest is reported
        assert np.array_equal(got_h[:, 0], want["hyps"][:, 0])
        same = int((got_h == want["hyps"]).all(axis=2).sum())
        # bounded, not free: most entries at the reference rank (measured 7 of 9), and whatever moved is the SAME
        # set of hypotheses per utterance in another order, with scores within the one-term noise
        n_all = got_h.shape[0] * got_h.shape[1]
        assert same >= (6 * n_all + 8) // 9, (same, n_all)
        for b in range(got_h.shape[0]):
            assert sorted(map(tuple, got_h[b].tolist())) == sorted(map(tuple, want["hyps"][b].tolist())), b
        assert score_err < 0.1, score_err
        print("bf16 N-best search: top-1 identical, %d of %d entries at the reference rank, max |score diff| %.2e"
              % (same, got_h.shape[0] * got_h.shape[1], score_err))
        if _raw is False:
            return None, None, None
    else:
        assert np.array_equal(got_h, want["hyps"])
        assert score_err < 2e-3, score_err
    net.train()
    net.zero_grad()
    if graphed:
        # the training half as ONE hipGraph (pika_amd.mbr.GraphedMbrStep): an eager warm-up call, the capture, two replays;
        # dropout is off, so every call leaves the same gradients -- the LAST replay's are compared
        step = mbr.GraphedMbrStep(net, rnnt_scale=rnnt_scale, sm_scale=sm, blk=blk, min_seen=1, warmup=1)
        for _ in range(4):
            net.zero_grad(set_to_none=True)
            step(data, target, len_b.int(), ali.int(), hyps, scores)
        assert step.broken is None, step.broken
        assert step.stats == {"replays": 3, "captures": 1, "eager": 1, "evictions": 0}, step.stats
        grads = {"g%03d" % i: (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().numpy()
                 for i, p in enumerate(net.parameters())}
        step.close()
        if _raw == "names":
            return M, dict(M.compact(grads), n=np.array(len(grads)), names=[k for k, _ in net.named_parameters()]), want
        return M, dict(M.compact(grads), n=np.array(len(grads))), want
    enc = net.encoder(data)                                                       # :124-138
    sos = torch.zeros(3, 1, dtype=torch.long, device=device)
    pred = net.predict(torch.cat((sos, target), dim=1))
    lp = ops.joint(enc, pred, net.fc1, net.fc_gate, net.fc2, log_softmax=True)
    if device == "cpu":
        from oracle import rnnt as O
        costs, g = O.rnnt_loss(lp.detach().numpy(), target.int().numpy(), len_b.int().numpy(), ali.int().numpy(),
                               dtype=np.float32)
        lp.backward(rnnt_scale * torch.from_numpy(g), retain_graph=True)          # the checker stands in on CPU
    else:
        from pika_amd.rnnt import RNNTLoss
        (rnnt_scale * RNNTLoss(blank=0).apply(lp, target.int(), len_b.int(), ali.int())).sum().backward(retain_graph=True)
    part = os.environ.get("PIKA_MBR_DEBUG_PART", "both") if _raw is True else "both"      # tools/mbr_mode_diff.py
    if part == "risk":
        net.zero_grad()
    if part != "rnnt":
        prob, dist, seq_grad, nonblk = mbr.risk_terms(hyps, scores, target, ali, blk, enc.device)
        mbr.mbr_backward(net, enc, hyps, seq_grad, nonblk, blk, sm)
    grads = {"g%03d" % i: (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().numpy()
             for i, p in enumerate(net.parameters())}
    if _raw == "names":
        return M, dict(M.compact(grads), n=np.array(len(grads)), names=[k for k, _ in net.named_parameters()]), want
    if _raw:
        return M, {"names": [k for k, _ in net.named_parameters()], "grads": list(grads.values())}, want
    got = dict(M.compact(grads), n=np.array(len(grads)))
    return M, got, want


def test_cpu_native_mbr_step_matches_the_reference_script_golden():
    M, got, want = _native_step_vs_script_golden("cpu")
    M.compare(got, want, rel=1e-3)


@pytest.mark.gpu
def test_gpu_native_mbr_step_matches_the_reference_script_golden(hip_device):
    from pika_amd import gemm as G
    old, G.PRECISION = G.PRECISION, "fp32"
    try:
        M, got, want = _native_step_vs_script_golden(hip_device)
    finally:
        G.PRECISION = old
    M.compare(got, want, rel=2e-3)


@pytest.mark.gpu
def test_gpu_graphed_mbr_training_half_matches_the_reference_script_golden(hip_device):
    """The training half of the step as ONE replayed hipGraph (GraphedMbrStep: one backward pass over rnnt + surrogate
    instead of the script's two) in the exact arithmetic: the gradients the unchanged reference script left in .grad
    (train_transducer_mbr_bmuf_otfaug.py:120-235), to the same 2e-3 as the eager native step."""
    from pika_amd import gemm as G
    old, G.PRECISION = G.PRECISION, "fp32"
    try:
        M, got, want = _native_step_vs_script_golden(hip_device, graphed=True)
    finally:
        G.PRECISION = old
    M.compare(got, want, rel=2e-3)


def _worst(got, want, names):
    """(worst sample error, worst L2-norm difference) over the parameters of two compact dumps, each relative to the
    parameter's gradient scale: its own max |g| -- or, for a bias, its weight's if that is larger.  A bias whose gradient
    is mathematically ZERO (the key projection's: softmax is invariant to a constant added to every score of a query; a
    bias in front of BatchNorm) holds only rounding noise in any arithmetic (exact mode: 3e-8 on this fixture; bf16
    backward products: 1.5e-3, i.e. 1.5 % of the key WEIGHT's gradient), so its own scale says nothing."""
    scale = {}
    for i, n in enumerate(names):
        scale[n] = float(want["m%03d" % i][2])
    ws, wn, who = 0.0, 0.0, None
    for i, n in enumerate(names):
        k = "%03d" % i
        ms, mg = want["m" + k], got["m" + k]
        sc = max(scale[n], scale.get(n[:-4] + "weight", 0.0) if n.endswith(".bias") else 0.0)
        if sc < 1e-12:
            continue
        e = float(np.abs(got["s" + k] - want["s" + k]).max() / sc)
        if e > ws:
            ws, who = e, n
        wn = max(wn, float(abs(mg[0] - ms[0]) / (ms[0] * sc / scale[n])) if scale[n] > 0 else 0.0)
    return ws, wn, who


@pytest.mark.gpu
def test_gpu_native_mbr_step_in_the_benchmarked_arithmetic(hip_device):
    """The same step in the arithmetic bench.py's MBR leg runs: the PACKAGE-DEFAULT product mode ("mixed": two-term
    forward, bf16 backward) with the N-best search in the default decode arithmetic (two fp16 terms) -- against the
    golden of the unchanged reference script.  The N-best (hypotheses, blanks included) must be the script's, scores
    within 2e-3.  Gradients carry the bf16 rounding of every backward product (2^-9 per operand): measured on MI355X,
    8e-2 of the parameter's largest entry on the prediction net's query / key projections (softmax backward: d(scores)
    rows sum to zero, so their bf16 rounding is a relative error of the small remainder), <= 2e-2 elsewhere; held to
    0.15 / 0.15 (the printed line has the measured values and the worst parameter).  The one-term bf16 search is run as
    well: on this fixture it must return the same N-best (NOT what bench.py times)."""
    from pika_amd import gemm as G
    assert G.PRECISION == "mixed"
    M, got, want = _native_step_vs_script_golden(hip_device, _raw="names")
    ws, wn, who = _worst(got, want, got["names"])
    print("MBR step, default arithmetic: N-best identical; worst gradient sample error %.2e of the parameter's scale "
          "(%s), worst parameter-norm difference %.2e" % (ws, who, wn))
    assert ws < 0.15 and wn < 0.15, (ws, wn, who)
    _native_step_vs_script_golden(hip_device, search_precision="bf16")   # asserts the N-best inside
    # ... and as bench.py's MBR leg runs the training half: ONE replayed hipGraph (pika_amd.mbr.GraphedMbrStep)
    M, got, want = _native_step_vs_script_golden(hip_device, _raw="names", graphed=True)
    ws, wn, who = _worst(got, want, got["names"])
    print("MBR step, default arithmetic, graphed training half: worst gradient sample error %.2e of the parameter's scale "
          "(%s), worst parameter-norm difference %.2e" % (ws, who, wn))
    assert ws < 0.15 and wn < 0.15, (ws, wn, who)


def test_edit_distances_library_call_equals_the_python_dp():
    """pika_edit_distances (host code of libpika_amd.so, all N-best pairs of a batch in one call) against the plain
    dynamic programme and hand cases -- what editdistance.eval returns (requirements.txt:1 of the reference)."""
    import random
    from pika_amd import mbr
    assert mbr.edit_distances([([1, 2, 3], [1, 2, 3]), ([], [4, 5]), ([7], []), ([1, 2, 3, 4], [2, 3, 5]), ([], [])]) == \
        [0, 2, 1, 2, 0]
    random.seed(3)
    pairs = [([random.randrange(6) for _ in range(random.randrange(0, 40))],
              [random.randrange(6) for _ in range(random.randrange(0, 25))]) for _ in range(200)]
    assert mbr.edit_distances(pairs) == [mbr.edit_distance(a, b) for a, b in pairs]
    assert mbr.edit_distances([]) == []


@pytest.mark.gpu
def test_gpu_graphed_mbr_step_replays_on_other_nbest_lists_and_buckets(hip_device):
    """GraphedMbrStep beyond the fixed batch of the golden: ONE captured graph must serve other N-best lists of its (S, U)
    bucket -- other symbols, lengths, scores, targets: everything travels through the static inputs -- a list outside the
    bucket gets its own graph, and every call leaves the gradients of the eager step (the script's two backward passes,
    `eager_step`) on the same weights.  Exact arithmetic; dropout off."""
    import copy
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from types import SimpleNamespace as NS
    from oracle.pika_ref import seeded_state_dict
    from model.transducer import Net
    from pika_amd import gemm as G
    from pika_amd import mbr
    V, B, beam, T_in, U = 120, 3, 3, 150, 6
    opt = NS(rnn_size=64, local_rank=0, decoder_type="transformer", brnn=False, encoder_type="tdnn", dropout=0.0,
             enc_layers=2, dec_layers=1, embd_dim=32, padding_idx=V)
    net = Net(opt, 240, V)
    net.encoder = type(net.encoder)(240, 0, 64, tdnn_nhid=64, tdnn_layers=6)
    net.pack_seq = False
    net.load_state_dict(seeded_state_dict(net, 77))
    net = net.to(hip_device).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    g = torch.Generator().manual_seed(5)
    Tp = (T_in - 24 + 3) // 4
    old, G.PRECISION = G.PRECISION, "fp32"
    step = mbr.GraphedMbrStep(net, rnnt_scale=0.1, sm_scale=0.9, blk=0, min_seen=1, warmup=1, s_bucket=32, u_bucket=8)
    ref_step = mbr.GraphedMbrStep(copy.deepcopy(net), rnnt_scale=0.1, sm_scale=0.9, blk=0)

    def batch(n_lab, seed):
        gg = torch.Generator().manual_seed(seed)
        feats = torch.randn(B, T_in, 240, generator=gg).to(hip_device)
        labels = torch.randint(1, V, (B, U), generator=gg)
        ali = torch.tensor([U, U - 1, U - 2], dtype=torch.int32)
        for b in range(B):
            labels[b, int(ali[b]):] = V
        x_len = torch.full((B,), Tp, dtype=torch.int32)
        hyps, scores = [], []
        for b in range(B):
            row, sc = [], []
            for j in range(beam):
                nl = max(1, n_lab - j - b)                      # labels of this hypothesis
                nb = Tp - int(torch.randint(0, 3, (1,), generator=gg))
                sym = [0] * nb + [int(v) for v in torch.randint(1, V, (nl,), generator=gg)]
                perm = torch.randperm(len(sym), generator=gg).tolist()
                row.append([torch.tensor(sym[i]) for i in perm])
                sc.append(torch.tensor(-1.0 - 0.3 * j - 0.01 * float(torch.rand(1, generator=gg))))
            hyps.append(row)
            scores.append(sc)
        return feats, labels.to(hip_device), x_len.to(hip_device), ali.to(hip_device), hyps, scores

    def grads_of(fn, args):
        fn.model.zero_grad(set_to_none=True)
        out = fn(*args) if fn is step else fn.eager_step(*args)
        torch.cuda.synchronize()
        return float(out), [None if p.grad is None else p.grad.detach().clone() for p in fn.model.parameters()]
    try:
        # same bucket three times (S = Tp + ~5..8 -> 64; U <= 8), then a longer label count (U bucket 16): a second graph
        for n_lab, seed in ((5, 1), (7, 2), (4, 3), (12, 4), (6, 5)):
            args = batch(n_lab, seed)
            got_loss, got = grads_of(step, args)
            want_loss, want = grads_of(ref_step, args)
            assert abs(got_loss - want_loss) <= 1e-4 * abs(want_loss) + 1e-5, (n_lab, got_loss, want_loss)
            for (name, _), a, b in zip(net.named_parameters(), got, want):
                if b is None:
                    assert a is None or float(a.abs().max()) == 0.0, name
                    continue
                scale = float(b.abs().max()) + 1e-12
                assert a is not None and float((a - b).abs().max()) <= 2e-3 * scale + 1e-7, (n_lab, name, float((a - b).abs().max()), scale)
        assert step.broken is None, step.broken
        # call 1 eager (warm-up), call 2 captures the first bucket, 3 replays it, 4 captures the wider label bucket, 5 is back on the first
        assert step.stats["captures"] == 2 and step.stats["replays"] == 4 and step.stats["eager"] == 1, step.stats
        assert len(step.entries) == 2
    finally:
        G.PRECISION = old
        step.close()


@pytest.mark.gpu
def test_gpu_graphed_mbr_step_on_batches_whose_shapes_never_recur(hip_device):
    """A corpus whose frame and label counts differ from batch to batch: no exact shape repeats, so a bucket of 64 frames x 8
    labels that has shown two shapes gets a graph at its upper boundary (192 frames, 8 labels) which the later batches of
    the bucket ride -- padding frames masked, labels padded -- and every call leaves the gradients of the eager step
    (`eager_step`) on the same weights.  Exact arithmetic; dropout off."""
    import copy
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from types import SimpleNamespace as NS
    from oracle.pika_ref import seeded_state_dict
    from model.transducer import Net
    from pika_amd import gemm as G
    from pika_amd import mbr
    V, B, beam, T_in, U = 120, 3, 3, 150, 6
    opt = NS(rnn_size=64, local_rank=0, decoder_type="transformer", brnn=False, encoder_type="tdnn", dropout=0.0,
             enc_layers=2, dec_layers=1, embd_dim=32, padding_idx=V)
    net = Net(opt, 240, V)
    net.encoder = type(net.encoder)(240, 0, 64, tdnn_nhid=64, tdnn_layers=6)
    net.pack_seq = False
    net.load_state_dict(seeded_state_dict(net, 77))
    net = net.to(hip_device).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    g = torch.Generator().manual_seed(5)
    old, G.PRECISION = G.PRECISION, "fp32"
    step = mbr.GraphedMbrStep(net, rnnt_scale=0.1, sm_scale=0.9, blk=0, min_seen=2, warmup=1, s_bucket=64, u_bucket=16)
    ref_step = mbr.GraphedMbrStep(copy.deepcopy(net), rnnt_scale=0.1, sm_scale=0.9, blk=0)

    def batch(n_lab, seed, T_in, U):
        Tp = (T_in - 24 + 3) // 4
        gg = torch.Generator().manual_seed(seed)
        feats = torch.randn(B, T_in, 240, generator=gg).to(hip_device)
        labels = torch.randint(1, V, (B, U), generator=gg)
        ali = torch.tensor([U, U - 1, max(U - 2, 1)], dtype=torch.int32)
        for b in range(B):
            labels[b, int(ali[b]):] = V
        x_len = torch.full((B,), Tp, dtype=torch.int32)
        hyps, scores = [], []
        for b in range(B):
            row, sc = [], []
            for j in range(beam):
                nl = max(1, n_lab - j - b)                      # labels of this hypothesis
                nb = Tp - int(torch.randint(0, 3, (1,), generator=gg))
                sym = [0] * nb + [int(v) for v in torch.randint(1, V, (nl,), generator=gg)]
                perm = torch.randperm(len(sym), generator=gg).tolist()
                row.append([torch.tensor(sym[i]) for i in perm])
                sc.append(torch.tensor(-1.0 - 0.3 * j - 0.01 * float(torch.rand(1, generator=gg))))
            hyps.append(row)
            scores.append(sc)
        return feats, labels.to(hip_device), x_len.to(hip_device), ali.to(hip_device), hyps, scores

    def grads_of(fn, args):
        fn.model.zero_grad(set_to_none=True)
        out = fn(*args) if fn is step else fn.eager_step(*args)
        torch.cuda.synchronize()
        return float(out), [None if p.grad is None else p.grad.detach().clone() for p in fn.model.parameters()]
    try:
        # frames / labels per batch; S <= 64 and hypothesis labels <= 16 throughout: one (S, U) bucket
        shapes = ((150, 6), (141, 5), (163, 7), (172, 6), (134, 4), (190, 8))
        for k, (T_in_k, U_k) in enumerate(shapes):
            args = batch(5 + k % 3, 10 + k, T_in_k, U_k)
            got_loss, got = grads_of(step, args)
            want_loss, want = grads_of(ref_step, args)
            assert abs(got_loss - want_loss) <= 1e-4 * abs(want_loss) + 1e-5, (k, got_loss, want_loss)
            big = max(float(b.norm() / b.numel() ** 0.5) for b in want if b is not None)
            for (name, _), a, b in zip(net.named_parameters(), got, want):
                if b is None:
                    assert a is None or float(a.abs().max()) == 0.0, name
                    continue
                scale = float(b.abs().max()) + 1e-12
                d = float((a - b).abs().max())
                if k < 2:           # the eager launch sequence on the batch as it is
                    assert d <= 2e-3 * scale + 1e-7, (k, name, d, scale)
                    continue
                # a padded time axis sums the BatchNorm statistics in another row order: single pre-activations within 1e-7 of
                # zero land on the other side of a ReLU (tests/test_train_step_gpu.py::test_padded_time_axis_*: the same bounds)
                assert d <= 0.3 * scale + 1e-7, (k, name, d, scale)
                if float(b.norm() / b.numel() ** 0.5) > 1e-4 * big:
                    assert float((a - b).norm() / b.norm()) < 0.15, (k, name, float((a - b).norm() / b.norm()))      # (measured: <= 3.8e-2)
        assert step.broken is None, step.broken
        # call 1 warm-up, 2 eager (first shape of bucket (192, 8) after the warm-up), 3 captures the bucket at (192, 8), 4-6 ride it
        assert step.stats.get("bucket_captures") == 1 and step.stats["captures"] == 1, step.stats
        assert step.stats["replays"] == 4 and step.stats["eager"] == 2 and step.stats.get("padded") == 3, step.stats
        assert [(k_[0][1], k_[2][1]) for k_ in step.entries] == [(192, 8)]
    finally:
        G.PRECISION = old
        step.close()
        ref_step.close()
