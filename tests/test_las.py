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
def test_gpu_las_rescore_matches_reference(hip_device):
    from pika_amd import gemm as G
    old, G.PRECISION = G.PRECISION, "fp32"
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
