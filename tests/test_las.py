"""LAS rescoring: per-token log-probs vs golden recorded from the REFERENCE
las.Net + TransducerDecoder.las_rescore (tests/golden/make_las_golden.py).  `bilas_rescore`
cannot be pinned: the reference calls its model with 7 positional arguments
(transducer_decoder.py:247-248) and no class in the reference tree accepts them."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import las_common as LC  # noqa: E402
from oracle.pika_ref import seeded_state_dict  # noqa: E402

GOLD = os.path.join(HERE, "golden", "las_rescore.npz")


def run(device):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from trainer.model import las                       # pickle/import path of a trained rescorer
    from decoder.transducer_decoder import TransducerDecoder
    z = np.load(GOLD)
    for attn in ("mlp", "general"):
        net = las.Net(LC.opt(attn), LC.C_IN, LC.V, LC.PAD)
        net.load_state_dict(seeded_state_dict(net, 31, scale=0.3))   # same keys/shapes as the reference
        net = net.eval().to(device)
        args = SimpleNamespace(las_rescorer=net, las_rescorer_bw=net, bilas_rescorer=None)
        d = TransducerDecoder(None, 1, 1, args=args)
        x, hyps = LC.inputs()
        x = x.to(device)
        for i, h in enumerate(hyps):
            tgt = torch.LongTensor([LC.SOS] + h + [LC.EOS]).to(device).unsqueeze(-1).unsqueeze(-1)
            got = d.las_rescore(x, tgt)
            assert np.allclose(got, z["%s/las/%d" % (attn, i)], rtol=1e-4, atol=1e-4), (attn, i)
            assert np.allclose(d.las_rescore(x, tgt, bw=True), got)
        # batched n-best scoring == one-by-one scoring
        batched = net.score_nbest(x, hyps, LC.SOS, LC.EOS)
        for i in range(len(hyps)):
            assert np.allclose(batched[i], z["%s/las/%d" % (attn, i)], rtol=1e-4, atol=1e-4)


def test_batched_nbest_scoring_equals_per_utterance_scoring():
    """score_nbest_batch (one encoder pass + one decoder pass for a whole ragged decode batch) returns what
    score_nbest returns utterance by utterance."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from trainer.model import las
    net = las.Net(LC.opt("mlp"), LC.C_IN, LC.V, LC.PAD)
    net.load_state_dict(seeded_state_dict(net, 31, scale=0.3))
    net = net.eval()
    g = torch.Generator().manual_seed(5)
    lens = [17, 23, 9]
    src = torch.zeros(23, 3, LC.C_IN)
    for b, n in enumerate(lens):
        src[:n, b] = torch.randn(n, LC.C_IN, generator=g)
    hyps = [[[3, 7, 7, 12], [5], []], [[8, 1, 30, 2, 2, 19, 4], [2, 2]], [[11]]]
    got = net.score_nbest_batch(src, lens, hyps, LC.SOS, LC.EOS)
    for b, n in enumerate(lens):
        want = net.score_nbest(src[:n, b:b + 1], hyps[b], LC.SOS, LC.EOS)
        assert len(got[b]) == len(want)
        for a, w in zip(got[b], want):
            assert np.allclose(a, w, rtol=1e-5, atol=1e-5)


def test_cpu_las_rescore_matches_reference():
    run("cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "default"])
def test_gpu_las_rescore_matches_reference(hip_device, precision):
    """In the exact mode and in the package default (what a decode script gets; the full-width scenario with the
    persistent BLSTM kernel on its path: tests/test_las_full.py)."""
    from pika_amd import gemm as G
    old = G.PRECISION
    if precision != "default":
        G.PRECISION = precision
    try:
        run(hip_device)
    finally:
        G.PRECISION = old


def _las_training_step(device):
    """One TRAINING step of the LAS model with the calling convention of train_las_bmuf_otfaug.py:227-239 and the
    decoder cross-entropy of its LASLossCompute: decoder outputs, loss and EVERY parameter gradient equal the
    reference's (golden recorded from trainer/model/las.py by tests/golden/make_las_train_golden.py)."""
    import torch.nn.functional as F
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from model import las                     # what importlib.import_module("model." + nnet_proto) resolves to
    z = np.load(os.path.join(HERE, "golden", "las_train.npz"))
    for attn in ("mlp", "general"):
        net = las.Net(LC.opt(attn), LC.C_IN, LC.V, LC.PAD)
        net.load_state_dict(seeded_state_dict(net, 31, scale=0.3))
        net = net.to(device).train()
        src, tgt, lens = LC.train_batch()
        src, tgt = src.to(device), tgt.to(device)
        outputs, _, _, enc_out = net.forward(src, tgt, lens, None, True, True)
        assert np.allclose(outputs.detach().cpu().numpy(), z["%s/outputs" % attn], rtol=1e-4, atol=1e-5)
        assert np.allclose(enc_out.detach().cpu().numpy(), z["%s/enc_out" % attn], rtol=1e-4, atol=1e-5)
        logp = F.log_softmax(net.dec_proj(outputs.view(-1, outputs.size(2))), dim=1)
        loss = F.nll_loss(logp, tgt[1:].contiguous().view(-1), ignore_index=LC.PAD, reduction="sum")
        loss.backward()
        assert abs(loss.item() - float(z["%s/loss" % attn])) < 1e-4 * abs(float(z["%s/loss" % attn]))
        n = 0
        for k, p in net.named_parameters():
            want = z["%s/grad/%s" % (attn, k)]
            got = (p.grad if p.grad is not None else torch.zeros_like(p)).cpu().numpy()
            assert np.allclose(got, want, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(want).max())), (attn, k)
            n += 1
        assert n == len([k for k in z.files if k.startswith(attn + "/grad/")])


def _las_training_modes(device):
    """The calls trainer/train_las_bmuf_otfaug.py makes besides the default one (:193-239): decoder pre-training
    (--pretrain_decoder: enable_enc False), encoder-only training on the script's CTC branch (dec_loss_scale 0: enable_dec
    False; :98-131), and a call that continues from the decoder state a previous call returned -- outputs, loss and every
    parameter gradient against the reference's (tests/golden/make_las_train_golden.py)."""
    import torch.nn.functional as F
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from model import las                     # what importlib.import_module("model." + nnet_proto) resolves to
    z = np.load(os.path.join(HERE, "golden", "las_train.npz"))

    def fresh():
        net = las.Net(LC.opt("mlp"), LC.C_IN, LC.V, LC.PAD)
        net.load_state_dict(seeded_state_dict(net, 31, scale=0.3))
        return net.to(device).train()

    def check(tag, net, loss, extra):
        loss.backward()
        want = float(z["%s/loss" % tag])
        assert abs(loss.item() - want) < 1e-4 * max(1.0, abs(want)), (tag, loss.item(), want)
        for k, v in extra.items():
            assert np.allclose(v.detach().cpu().numpy(), z["%s/%s" % (tag, k)], rtol=1e-4, atol=1e-5), (tag, k)
        n = 0
        for k, p in net.named_parameters():
            w = z["%s/grad/%s" % (tag, k)]
            got = (p.grad if p.grad is not None else torch.zeros_like(p)).cpu().numpy()
            assert np.allclose(got, w, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(w).max())), (tag, k)
            n += 1
        assert n == len([k for k in z.files if k.startswith(tag + "/grad/")])
    src, tgt, lens = LC.train_batch()
    src, tgt = src.to(device), tgt.to(device)
    net = fresh()
    outputs, a, b, c = net.forward(src, tgt, lens, None, True, False)
    assert a is None and b is None and c is None
    logp = F.log_softmax(net.dec_proj(outputs.view(-1, outputs.size(2))), dim=1)
    check("pretrain_dec", net, F.nll_loss(logp, tgt[1:].contiguous().view(-1), ignore_index=LC.PAD, reduction="sum"),
          {"outputs": outputs})
    net = fresh()
    o, a, b, enc_out = net.forward(src, tgt, lens, None, False, True)
    assert o is None and a is None and b is None
    L, B_, _ = enc_out.shape
    pout = net.enc_proj(enc_out.view(-1, enc_out.size(2))).view(L, B_, -1)
    t2 = tgt.view(tgt.size(0), -1).transpose(0, 1)
    mask = torch.lt(t2, LC.PAD) & torch.gt(t2, 1)
    ctc = torch.nn.CTCLoss()(pout.cpu() if device == "cpu" else pout, t2[mask].int().cpu(), torch.as_tensor(lens).int(),
                             mask.int().sum(1).cpu())
    check("enc_only_ctc", net, ctc, {"enc_out": enc_out})
    net = fresh()
    half = (tgt.size(0) - 1) // 2
    tgt_a, tgt_b = tgt[:half + 1], tgt[half:]
    out_a, _, st, _ = net.forward(src, tgt_a, lens, None, True, True)
    out_b, _, _, _ = net.forward(src, tgt_b, lens, st, True, True)
    outputs = torch.cat([out_a, out_b], 0)
    logp = F.log_softmax(net.dec_proj(outputs.view(-1, outputs.size(2))), dim=1)
    tg = torch.cat([tgt_a[1:], tgt_b[1:]], 0)
    check("carried_state", net, F.nll_loss(logp, tg.contiguous().view(-1), ignore_index=LC.PAD, reduction="sum"),
          {"outputs": outputs})


def test_las_training_modes_match_reference():
    _las_training_modes("cpu")


@pytest.mark.gpu
def test_gpu_las_training_modes_match_reference(hip_device):
    from pika_amd import gemm as G
    old, G.PRECISION = G.PRECISION, "fp32"
    try:
        _las_training_modes(hip_device)
    finally:
        G.PRECISION = old


def test_las_training_step_matches_reference():
    _las_training_step("cpu")


@pytest.mark.gpu
def test_gpu_las_training_step_matches_reference(hip_device):
    """The same step on the MI355X (Linear layers on the MFMA GEMM in its fp32-exact mode, LSTMs on MIOpen): every
    parameter gradient against the golden recorded from the reference's trainer/model/las.py."""
    from pika_amd import gemm as G
    old, G.PRECISION = G.PRECISION, "fp32"
    try:
        _las_training_step(hip_device)
    finally:
        G.PRECISION = old


def _script_loop(device, batch_ahead, monkeypatch):
    """The rescoring part of decode_transducer.py:130-156 on a tiny transducer + tiny forward / backward rescorers: decode
    a batch, then ask for the scores one (utterance, n-best entry, direction) at a time."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    import decode_common as D
    from test_decode import build
    from trainer.model import las
    from decoder.transducer_decoder import TransducerDecoder
    from decoder.beam_transducer import GlobalScorer
    monkeypatch.setenv("PIKA_LAS_BATCH_AHEAD", "1" if batch_ahead else "0")
    net = build("transformer", device)
    x, x_len = D.inputs()
    x, x_len = x.to(device), x_len.to(device)
    C_enc = net.hid_dim
    V = net.output_dim
    sos, eos, pad = V, V + 1, V + 2
    rescorers = []
    for seed in (31, 32):
        r = las.Net(LC.opt("mlp"), C_enc, V + 2, pad)
        r.load_state_dict(seeded_state_dict(r, seed, scale=0.3))
        rescorers.append(r.eval().to(device))
    args = SimpleNamespace(las_rescorer=rescorers[0], las_rescorer_bw=rescorers[1], bilas_rescorer=None, nonblk_reward=0.0)
    d = TransducerDecoder(net, batch_size=x.shape[0], beam_size=4, n_best=4, blk=0, global_scorer=GlobalScorer(),
                          sm_scale=0.8, cuda=(device != "cpu"), beam_prune=True, args=args)
    ret, enc_out = d.decode_batch(x, x_len, [int(v) + 100 for v in x_len])
    out = []
    for i in range(x.shape[0]):
        for j in range(4):
            hyp = [e.item() for e in ret["predictions"][i][j] if e != 0]
            las_in = enc_out[i].unsqueeze(1)
            tgt = torch.LongTensor([sos] + hyp + [eos]).to(device).unsqueeze(-1).unsqueeze(-1)
            fw = d.las_rescore(las_in, tgt)
            tgt = torch.LongTensor([sos] + hyp[::-1] + [eos]).to(device).unsqueeze(-1).unsqueeze(-1)
            bw = d.las_rescore(las_in, tgt, bw=True)
            assert len(fw) == len(bw) == len(hyp) + 1
            out.append((fw, bw))
    # a question that is not about the last batch falls back to scoring on its own
    other = torch.randn_like(enc_out[0]).unsqueeze(1)
    tgt = torch.LongTensor([sos, 3, 4, eos]).to(device).unsqueeze(-1).unsqueeze(-1)
    assert len(d.las_rescore(other, tgt)) == 3
    return out, (d._nbest["scores"] if batch_ahead else None)


def test_batch_ahead_rescoring_answers_the_script_loop_with_the_per_hypothesis_values(monkeypatch):
    """The unchanged decode script asks for LAS scores one hypothesis at a time; the decoder answers from ONE batched pass
    per rescorer over the n-best lists decode_batch returned (transducer_decoder._batch_ahead).  Same values as scoring
    every hypothesis on its own, on the CPU path."""
    ahead, tables = _script_loop("cpu", True, monkeypatch)
    alone, _ = _script_loop("cpu", False, monkeypatch)
    assert set(k[0] for k in tables) == {"fw", "bw"}            # one batched pass per rescorer, both were used
    for (f1, b1), (f2, b2) in zip(ahead, alone):
        assert np.allclose(f1, f2, rtol=1e-5, atol=1e-5) and np.allclose(b1, b2, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "default"])
def test_gpu_batch_ahead_rescoring_answers_the_script_loop(hip_device, monkeypatch, precision):
    from pika_amd import gemm as G
    old = G.PRECISION
    if precision != "default":
        G.PRECISION = precision
    try:
        ahead, tables = _script_loop(hip_device, True, monkeypatch)
        alone, _ = _script_loop(hip_device, False, monkeypatch)
    finally:
        G.PRECISION = old
    assert set(k[0] for k in tables) == {"fw", "bw"}
    for (f1, b1), (f2, b2) in zip(ahead, alone):
        assert np.allclose(f1, f2, rtol=1e-4, atol=1e-4) and np.allclose(b1, b2, rtol=1e-4, atol=1e-4)


def test_step_lists_and_one_blob_upload():
    """Host side of the fused token loop (InputFeedRNNDecoder.step_lists): per step the active rows, ordered by utterance,
    with each utterance's range in the step's list; and _h2d_many (one upload for many arrays) hands every array back
    unchanged."""
    from pika_amd.model.las import InputFeedRNNDecoder, _h2d_many
    rng = np.random.default_rng(5)
    B, N, L = 5, 37, 9
    own = np.sort(rng.integers(0, B, N))
    own[own == 3] = 2                                   # an utterance without rows
    own = own[rng.permutation(N)]
    first = rng.integers(0, 4, N)
    end = first + rng.integers(0, L - 3, N)
    hl = InputFeedRNNDecoder.step_lists(own, (first, end), L, N, B)
    assert hl["uoff"].shape == (L, B + 1) and hl["n_act"].shape == (L,)
    for t in range(L):
        rows = hl["qlist"][hl["qoff"][t]:hl["qoff"][t + 1]]
        assert sorted(rows.tolist()) == sorted(np.nonzero((first <= t) & (t < end))[0].tolist())
        assert len(rows) == hl["n_act"][t] and np.all(np.diff(own[rows]) >= 0)          # ordered by utterance
        for b in range(B):
            seg = rows[hl["uoff"][t, b]:hl["uoff"][t, b + 1]]
            assert np.all(own[seg] == b) and len(seg) == int((own[rows] == b).sum())
    arrays = {"a": rng.integers(0, 9, (3, 5)).astype(np.int64), "b": np.arange(7, dtype=np.int32), "c": np.zeros(0, np.int32),
              "d": rng.standard_normal(11).astype(np.float32)}
    up = _h2d_many(arrays, "cpu")
    for k, v in arrays.items():
        assert up[k].shape == v.shape and np.array_equal(up[k].numpy(), v)
