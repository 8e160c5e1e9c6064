"""Beam-search parity: our vectorised decoder vs golden n-best lists
recorded from the REFERENCE decoder on the same tiny model (tests/golden/make_decode_golden.py).
Hypotheses (blanks included, as the reference returns them) must be IDENTICAL; scores within
1e-4 (fp32 reassociation of the split joint)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import model_common as C  # noqa: E402
import decode_common as D  # noqa: E402
from oracle.pika_ref import seeded_state_dict  # noqa: E402

GOLD = os.path.join(HERE, "golden", "decode_tiny.npz")


def build(dec, device):
    from pika_amd.model import transducer, encoder
    net = C.build(transducer, encoder, dec)
    net.load_state_dict(seeded_state_dict(net, C.SEED))
    D.tweak(net)
    return net.eval().to(device)


def run_all(dec, device):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from decoder.transducer_decoder import TransducerDecoder   # the scripts' import path
    from decoder.beam_transducer import GlobalScorer
    z = np.load(GOLD)
    net = build(dec, device)
    x, x_len = D.inputs()
    x, x_len_d = x.to(device), x_len.to(device)
    for name, cfg in D.SCENARIOS.items():
        args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None,
                               nonblk_reward=0.0)
        d = TransducerDecoder(net, batch_size=x.shape[0], beam_size=cfg["beam"], n_best=cfg["n_best"],
                              blk=0, global_scorer=GlobalScorer(), sm_scale=cfg["sm_scale"],
                              cuda=(device != "cpu"), beam_prune=True, args=args)
        ret, enc = d.decode_batch(x, x_len_d, D.max_len(cfg, x_len))
        got = D.pack(ret["predictions"], ret["scores"])
        pre = "%s/%s/" % (dec, name)
        assert np.array_equal(got["lens"], z[pre + "lens"]), (dec, name, got["lens"], z[pre + "lens"])
        assert np.array_equal(got["hyps"], z[pre + "hyps"]), (dec, name)
        assert np.allclose(got["scores"], z[pre + "scores"], rtol=1e-5, atol=1e-4), (dec, name)
        # return types the decode script relies on (decode_transducer.py:139,166): 0-dim tensors
        e = ret["predictions"][0][0][0] if len(ret["predictions"][0][0]) else torch.tensor(0)
        assert hasattr(e, "item")


@pytest.mark.parametrize("dec", ["transformer", "rnn"])
def test_cpu_decoder_matches_reference_nbest(dec):
    run_all(dec, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("dec", ["transformer", "rnn"])
def test_gpu_decoder_matches_reference_nbest(hip_device, dec):
    from pika_amd import gemm as G
    old, G.PRECISION = G.PRECISION, "fp32"   # fp32-class joint: identical arg-max decisions
    try:
        run_all(dec, hip_device)
    finally:
        G.PRECISION = old


def test_incremental_prediction_net_equals_prefix_recompute():
    """The ancestry-cached one-position-per-step prediction net (decoder/prednet_cache.py) and the
    reference-style re-run over the whole prefix produce the same n-best lists and scores."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from decoder.transducer_decoder import TransducerDecoder
    from decoder.beam_transducer import GlobalScorer
    net = build("transformer", "cpu")
    x, x_len = D.inputs()
    outs = []
    for inc in (True, False):
        args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=0.0)
        d = TransducerDecoder(net, batch_size=4, beam_size=5, n_best=5, blk=0, global_scorer=GlobalScorer(),
                              sm_scale=0.9, cuda=False, beam_prune=True, args=args)
        d.incremental = inc
        ret, _ = d.decode_batch(x, x_len, [40] * 4)
        outs.append(D.pack(ret["predictions"], ret["scores"]))
        assert (d._inc is not None) == inc
    assert np.array_equal(outs[0]["hyps"], outs[1]["hyps"])
    assert np.allclose(outs[0]["scores"], outs[1]["scores"], atol=1e-4)


@pytest.mark.gpu
def test_gpu_fused_advance_kernel_equals_tensor_op_path(hip_device):
    """include/pika_decode.h vs the torch-op `_advance`, both on the GPU, several beam widths."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from decoder.transducer_decoder import TransducerDecoder
    from decoder.beam_transducer import GlobalScorer
    from pika_amd import gemm as G
    old, G.PRECISION = G.PRECISION, "fp32"
    try:
        for dec in ("rnn", "transformer"):
            net = build(dec, hip_device)
            x, x_len = D.inputs()
            x, xl = x.to(hip_device), x_len.to(hip_device)
            for beam, nb, ml in ((1, 1, None), (5, 3, None), (16, 16, None), (7, 7, 9)):
                outs = []
                for fused in (True, False):
                    args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=0.0)
                    d = TransducerDecoder(net, 4, beam, n_best=nb, blk=0, global_scorer=GlobalScorer(), sm_scale=0.85,
                                          cuda=True, beam_prune=True, args=args)
                    d.fused_step = fused
                    mlen = [int(v) + 100 for v in x_len] if ml is None else [ml] * 4
                    ret, _ = d.decode_batch(x, xl, mlen)
                    outs.append(D.pack(ret["predictions"], ret["scores"]))
                assert np.array_equal(outs[0]["hyps"], outs[1]["hyps"]), (dec, beam)
                assert np.allclose(outs[0]["scores"], outs[1]["scores"], atol=2e-4), (dec, beam)
    finally:
        G.PRECISION = old


@pytest.mark.gpu
@pytest.mark.parametrize("rows,L,d,heads", [(5, 1, 64, 4), (37, 17, 256, 4), (64, 48, 1024, 16), (9, 33, 2048, 8), (12, 40, 512, 8)])
def test_incremental_attention_kernel(hip_device, rows, L, d, heads):
    """pika_incremental_attention vs the gather / matmul / masked softmax chain of prednet_cache.py on random
    caches, ancestry lists and positions (keys at positions <= pos, the new position read from `node`)."""
    import math
    from pika_amd import _lib
    g = torch.Generator().manual_seed(rows * 7 + L)
    cap = rows * (L + 2) + 3
    Kc = torch.randn(cap, d, generator=g).to(hip_device)
    Vc = torch.randn(cap, d, generator=g).to(hip_device)
    q = torch.randn(rows, d, generator=g).to(hip_device)
    anc = torch.randint(0, cap - 1, (rows, L + 3), generator=g).to(hip_device)
    pos = torch.randint(0, L, (rows,), generator=g).to(hip_device)
    pos[0] = L - 1
    node = torch.randint(0, cap - 1, (rows,), generator=g).to(hip_device)
    out = torch.empty_like(q)
    rc = _lib.lib().pika_incremental_attention(q.data_ptr(), Kc.data_ptr(), Vc.data_ptr(), anc.data_ptr(), anc.stride(0),
                                               pos.data_ptr(), node.data_ptr(), rows, L, d, heads, out.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "pika_incremental_attention")
    dh = d // heads
    ar = torch.arange(L, device=hip_device).unsqueeze(0)
    anc_now = torch.where(ar.eq(pos.unsqueeze(1)), node.unsqueeze(1), anc[:, :L])
    valid = ar <= pos.unsqueeze(1)
    Kp = Kc.double().index_select(0, anc_now.reshape(-1)).view(rows, L, heads, dh)
    Vp = Vc.double().index_select(0, anc_now.reshape(-1)).view(rows, L, heads, dh)
    qh = (q.double() / math.sqrt(dh)).view(rows, heads, 1, dh)
    sc = torch.matmul(qh, Kp.permute(0, 2, 3, 1)).masked_fill(~valid.view(rows, 1, 1, L), -1e18)
    want = torch.matmul(torch.softmax(sc, dim=-1), Vp.permute(0, 2, 1, 3)).reshape(rows, d)
    assert (out.double() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


def _results_loop_form(self):
    """The extraction spelled out per utterance as beam_transducer.py:196-243 / transducer_decoder.py:204-217 do it (what
    BeamState.results was before it went to whole-array form): sort_finished(minimum=n_best) + get_hyp for every utterance
    (beam_transducer.py:196-243, transducer_decoder.py:204-217).  Host side: the search is
    over, only K-sized lists remain."""
    import numpy as np
    S = self.steps
    ys = self.ys_hist[:S + 1].cpu().numpy()
    ks = self.ks_hist[:S].cpu().numpy()
    fin_n = self.fin_n.cpu().clamp(max=self.fin_cap - 1).numpy()
    nmax = int(fin_n.max()) if self.B else 0
    fin_score, fin_step, fin_k = (t[:, :nmax].cpu().numpy() for t in
                                  (self.fin_score, self.fin_step, self.fin_k))
    scores = self.scores.cpu().numpy()
    B, nb = self.B, self.n_best
    sel_score = np.zeros((B, nb), np.float32)
    sel_step = np.zeros((B, nb), np.int64)
    sel_k = np.zeros((B, nb), np.int64)
    for b in range(B):
        n = int(fin_n[b])
        fin = [(fin_score[b, i], int(fin_step[b, i]), int(fin_k[b, i])) for i in range(n)]
        while len(fin) < nb:                                              # :202-210 (i stays 0)
            fin.append((scores[b, 0], S, 0))
        fin.sort(key=lambda a: -float(a[0]))                              # :212 (stable)
        for j, (s, t, k) in enumerate(fin[:nb]):
            sel_score[b, j], sel_step[b, j], sel_k[b, j] = s, t, k
    # get_hyp (:234-243) for all B*n_best entries at once: walk the back-pointers from each
    # entry's own finishing step down to 0
    smax = int(sel_step.max()) if B else 0
    out = np.full((B, nb, max(smax, 1)), self.blk, np.int64)
    kcur = sel_k.copy()
    brow = np.arange(B)[:, None]
    for j in range(smax - 1, -1, -1):
        act = j < sel_step
        out[:, :, j] = np.where(act, ys[j + 1][brow, kcur], out[:, :, j])
        kcur = np.where(act, ks[j][brow, kcur], kcur)
    preds, out_scores = [], []
    for b in range(B):
        # hyp[:-1]: strip the trailing eos (:214); elements expose .item() like the
        # reference's 0-dim tensors (decode_transducer.py:139)
        preds.append([list(out[b, j, :max(int(sel_step[b, j]) - 1, 0)]) for j in range(nb)])
        out_scores.append([torch.tensor(float(sel_score[b, j])) for j in range(nb)])
    return preds, out_scores


def _fake_search_state(seed, B=9, K=6, nb=6, S=57, V=40):
    """A finished search's device state with everything the extraction has to cope with: utterances with fewer finished
    entries than n_best (none at all, too), tied scores, entries that finished at the last step."""
    g = torch.Generator().manual_seed(seed)
    o = SimpleNamespace(steps=S, B=B, n_best=nb, blk=0, fin_cap=K * S + 1)
    o.ys_hist = torch.randint(0, V, (S + 3, B, K), generator=g)
    o.ks_hist = torch.randint(0, K, (S + 3, B, K), generator=g)
    o.fin_n = torch.randint(0, 4 * nb, (B,), generator=g)
    o.fin_n[0], o.fin_n[1], o.fin_n[2] = 0, 2, nb
    o.fin_score = -torch.rand(B, o.fin_cap, generator=g) * 30
    o.fin_score[3, 4] = o.fin_score[3, 1]
    o.fin_score[1, 0] = o.fin_score[1, 1]
    o.fin_step = torch.randint(1, S + 1, (B, o.fin_cap), generator=g)
    o.fin_step[4, :3] = S
    o.fin_k = torch.randint(0, K, (B, o.fin_cap), generator=g)
    o.scores = -torch.rand(B, K, generator=g) * 30
    o.scores[1, 0] = o.fin_score[1, 0]                          # a fill-up entry tied with a finished one
    return o


def test_results_extraction_equals_the_per_utterance_form():
    """BeamState.results (sorting, fill-ups and back-pointer walks on whole arrays) returns what the per-utterance loops
    return: same hypotheses (blanks included), same scores in the same order, elements with .item()."""
    from pika_amd.decoder.beam_search import BeamState
    for seed in range(6):
        o = _fake_search_state(seed)
        want_p, want_s = _results_loop_form(o)
        got_p, got_s = BeamState.results(o)
        assert len(got_p) == o.B and all(len(r) == o.n_best for r in got_p)
        for wr, gr in zip(want_p, got_p):
            for wh, gh in zip(wr, gr):
                assert isinstance(gh, list) and [int(e) for e in wh] == [e.item() for e in gh]
        for wr, gr in zip(want_s, got_s):
            assert [float(e) for e in wr] == [e.item() for e in gr]


@pytest.mark.gpu
@pytest.mark.parametrize("dec", ["transformer", "rnn"])
def test_gpu_launch_chain_at_the_recipes_vocabulary(hip_device, dec):
    """V = 6268 (egs/train_transducer_bmuf_otfaug.sh:37; beyond the 5120 the one-launch advance used to stop at): the launch
    chain of fused_step takes the search -- and returns the n-best lists of the tensor-op path."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from decoder.transducer_decoder import TransducerDecoder
    from decoder.beam_transducer import GlobalScorer
    from pika_amd import gemm as G
    from pika_amd.model import transducer, encoder
    V = 6268
    torch.manual_seed(0)
    opt = C.make_opt(dec)
    opt.padding_idx = V
    o2 = SimpleNamespace(**vars(opt))
    o2.encoder_type, o2.enc_layers = "rnn", 1
    net = transducer.Net(o2, C.D_IN, V)
    net.encoder = encoder.Net(C.D_IN, 0, C.H, tdnn_nhid=C.NHID, tdnn_layers=C.LAYERS)
    net.pack_seq = False
    net.load_state_dict(seeded_state_dict(net, C.SEED))
    D.tweak(net)
    net = net.eval().to(hip_device)
    old, G.PRECISION = G.PRECISION, "fp32"
    try:
        x, x_len = D.inputs()
        x, xl = x.to(hip_device), x_len.to(hip_device)
        for beam, nb in ((4, 4), (16, 16)):
            outs = []
            for chain in (True, False):
                args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=0.0)
                d = TransducerDecoder(net, 4, beam, n_best=nb, blk=0, global_scorer=GlobalScorer(), sm_scale=0.85, cuda=True,
                                      beam_prune=True, args=args)
                d.fused_step = chain
                ret, _ = d.decode_batch(x, xl, [int(v) + 100 for v in x_len])
                assert bool(d.timing.get("launches_per_step")) == chain, d.timing
                outs.append(D.pack(ret["predictions"], ret["scores"]))
            assert np.array_equal(outs[0]["hyps"], outs[1]["hyps"]), (dec, beam)
            assert np.allclose(outs[0]["scores"], outs[1]["scores"], atol=2e-4), (dec, beam)
    finally:
        G.PRECISION = old
