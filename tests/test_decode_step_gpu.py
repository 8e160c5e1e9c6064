"""Per-step decode kernels (include/pika_decode_step.h) against plain fp64 / torch formulas and against the
logits-materialising advance they replace (pika_beam_advance, itself pinned by the reference-decoder goldens of
tests/test_decode.py)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _st():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("terms,tol", [(1, 2e-2), (2, 2e-4), (3, 2e-6), (4, 4e-6)])
@pytest.mark.parametrize("M,N,K", [(70, 100, 64), (1024, 512, 2560), (33, 1536, 96), (1024, 2048, 512), (300, 4100, 160),
                                   (600, 2056, 96)])      # (the last three: the wide-tile kernel, 128 / 128 / 64 + 64 rows)
def test_dgemm_bias_relu_residual_scatter(hip_device, terms, tol, M, N, K):
    from pika_amd.decoder.fused_step import DGemm, PackedWeight, DG_RELU, DG_ROWMASK
    from pika_amd import _lib
    g = torch.Generator().manual_seed(M + N + K + terms)
    lda, ldc = K + 32, N + 8
    A = torch.randn(M, lda, generator=g).to(hip_device)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(hip_device)
    bias = torch.randn(N, generator=g).to(hip_device)
    res = torch.randn(M, N, generator=g).to(hip_device)
    node = torch.randperm(M + 5, generator=g)[:M].to(hip_device)
    skip = int(node[3])
    C = torch.full((M, ldc), 7.0, device=hip_device)
    C2 = torch.zeros(M + 5, N, device=hip_device)
    pw = PackedWeight(W, terms)
    Kp = (K + 31) // 32 * 32
    A[:, K:Kp] = 0
    d = DGemm()
    d.A, d.lda, d.W, d.bias, d.res, d.ldr = A.data_ptr(), lda, pw.buf.data_ptr(), bias.data_ptr(), res.data_ptr(), N
    d.C, d.ldc, d.C2, d.ldc2, d.node, d.skip_node = C.data_ptr(), ldc, C2.data_ptr(), N, node.data_ptr(), skip
    d.M, d.N, d.K, d.terms, d.flags = M, N, K, terms, DG_RELU | DG_ROWMASK
    _lib.check(_lib.lib().pika_dgemm(ctypes.byref(d), _st()), "pika_dgemm")
    want = (torch.relu(A[:, :K].double() @ W.double().t() + bias.double()) + res.double())
    got = C[:, :N].double()
    scale = want.abs().max().item()
    keep = node != skip
    assert (got[keep] - want[keep]).abs().max().item() <= tol * scale
    assert torch.all(C[~keep][:, :N] == 7.0) and torch.all(C[:, N:] == 7.0)        # masked row, padding untouched
    assert (C2[node[keep]].double() - want[keep]).abs().max().item() <= tol * scale
    # compact row lists: only the first *m_dev rows are computed, and they land at rows crow[r]
    md = torch.tensor([M - 7], dtype=torch.int32, device=hip_device)
    crow = torch.randperm(M, generator=g).to(hip_device)
    C.fill_(7.0)
    d.flags, d.C2, d.m_dev, d.crow = DG_RELU, None, md.data_ptr(), crow.data_ptr()
    _lib.check(_lib.lib().pika_dgemm(ctypes.byref(d), _st()), "pika_dgemm")
    got = C[:, :N].double()
    assert (got[crow[:M - 7]] - want[:M - 7]).abs().max().item() <= tol * scale
    assert torch.all(C[crow[M - 7:]][:, :N] == 7.0)
    # gather lists: launch row e stands for row rowlist[off + e] of A, res and C (LAS token loop: the step's active rows)
    n_e = max(M // 3, 1)
    off = 5
    rl = torch.cat([torch.full((off,), -1, dtype=torch.int32), torch.randperm(M, generator=g)[:n_e].to(torch.int32)])
    rl_d, off_d = rl.to(hip_device), torch.tensor([off], dtype=torch.int32, device=hip_device)
    ne_d = torch.tensor([n_e], dtype=torch.int32, device=hip_device)
    C.fill_(7.0)
    d.crow, d.m_dev, d.rowlist, d.rowoff_dev = None, ne_d.data_ptr(), rl_d.data_ptr(), off_d.data_ptr()
    _lib.check(_lib.lib().pika_dgemm(ctypes.byref(d), _st()), "pika_dgemm")
    rows = rl[off:].long().to(hip_device)
    got = C[:, :N].double()
    assert (got[rows] - want[rows]).abs().max().item() <= tol * scale
    rest = torch.ones(M, dtype=torch.bool, device=hip_device)
    rest[rows] = False
    assert torch.all(C[rest][:, :N] == 7.0)


@pytest.mark.parametrize("terms,tol", [(1, 2e-2), (3, 3e-6), (4, 5e-6)])
@pytest.mark.parametrize("M,N,K", [(170, 1536, 512), (1024, 512, 512), (37, 100, 96), (300, 1024, 1024)])
def test_dgemm_layer_norm_on_the_way_in(hip_device, terms, tol, M, N, K):
    """pika_dgemm with ln_gamma / ln_beta == LayerNorm (onmt LayerNorm, eps 1e-6, biased variance) followed by the product:
    the rows are normalised inside the launch (row statistics over the waves of a workgroup)."""
    from pika_amd.decoder.fused_step import DGemm, PackedWeight, DG_RELU
    from pika_amd import _lib
    g = torch.Generator().manual_seed(M + N + K + terms)
    A = (torch.randn(M, K, generator=g) * 3 + 0.7).to(hip_device)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(hip_device)
    bias = torch.randn(N, generator=g).to(hip_device)
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).to(hip_device)
    beta = (0.1 * torch.randn(K, generator=g)).to(hip_device)
    res = torch.randn(M, N, generator=g).to(hip_device)
    C = torch.full((M, N), 7.0, device=hip_device)
    md = torch.tensor([M - 3], dtype=torch.int32, device=hip_device)
    pw = PackedWeight(W, terms)
    d = DGemm()
    d.A, d.lda, d.W, d.bias, d.res, d.ldr = A.data_ptr(), K, pw.buf.data_ptr(), bias.data_ptr(), res.data_ptr(), N
    d.C, d.ldc, d.M, d.N, d.K, d.terms, d.flags = C.data_ptr(), N, M, N, K, terms, DG_RELU
    d.m_dev = md.data_ptr()
    d.ln_gamma, d.ln_beta, d.ln_eps = gamma.data_ptr(), beta.data_ptr(), 1e-6
    _lib.check(_lib.lib().pika_dgemm(ctypes.byref(d), _st()), "pika_dgemm ln")
    x = A.double()
    xn = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-6) * gamma.double() + beta.double()
    want = torch.relu(xn @ W.double().t() + bias.double()) + res.double()
    scale = want.abs().max().item()
    assert (C[:M - 3].double() - want[:M - 3]).abs().max().item() <= tol * scale
    assert torch.all(C[M - 3:] == 7.0)
    d.K = K - 4                                           # not a whole number of k-tiles: refused, not mis-computed
    assert _lib.lib().pika_dgemm(ctypes.byref(d), _st()) < 0


def test_dgemm_gate_epilogue(hip_device):
    """Prediction halves of fc1 / fc_gate + gathered encoder halves + tanh * sigmoid (transducer.py:107-109)."""
    from pika_amd.decoder.fused_step import DGemm, PackedWeight, DG_GATE
    from pika_amd import _lib
    g = torch.Generator().manual_seed(5)
    B, beam, T, H = 3, 4, 11, 64
    R = B * beam
    state = torch.randn(R, H, generator=g).to(hip_device)
    wp = (torch.randn(2 * H, H, generator=g) * 0.2).to(hip_device)
    e_all = torch.randn(B * T, 2 * H, generator=g).to(hip_device)
    t_idx = torch.randint(-1, T + 2, (R,), generator=g).to(hip_device)
    pw = PackedWeight(wp, 3, interleave2=True)
    out = torch.empty(R, H, device=hip_device)
    d = DGemm()
    d.A, d.lda, d.W, d.C, d.ldc = state.data_ptr(), H, pw.buf.data_ptr(), out.data_ptr(), H
    d.e_all, d.t_idx, d.T, d.beam = e_all.data_ptr(), t_idx.data_ptr(), T, beam
    d.M, d.N, d.K, d.terms, d.flags = R, 2 * H, H, 3, DG_GATE
    _lib.check(_lib.lib().pika_dgemm(ctypes.byref(d), _st()), "pika_dgemm gate")
    z = state.double() @ wp.double().t()
    rows = torch.arange(R, device=hip_device) // beam * T + t_idx.clamp(0, T - 1)
    e = e_all[rows].double()
    want = torch.tanh(z[:, :H] + e[:, :H]) * torch.sigmoid(z[:, H:] + e[:, H:])
    assert (out.double() - want).abs().max().item() < 2e-6


def _beam_inputs(B, K, V, L, g, dev, first):
    s = dict(scores=torch.randn(B, K, generator=g), lm_scores=torch.zeros(B, K),
             y=torch.randint(0, V, (B, K), generator=g), t_idx=torch.randint(0, 6, (B, K), generator=g),
             num_frames=torch.full((B,), 6, dtype=torch.long), max_len=torch.full((B,), 30, dtype=torch.long),
             hyp=torch.randint(1, V, (B, K, L), generator=g), hyp_len=torch.randint(0, 5, (B, K), generator=g),
             eos=torch.zeros(B, dtype=torch.uint8))
    s["y"][0, 1] = -1                                   # an eos row
    s["hyp"][1, 2] = s["hyp"][1, 0]
    s["hyp_len"][1, 2] = s["hyp_len"][1, 0] = 3         # a duplicate partial hypothesis
    s["y"][1, 0] = s["y"][1, 2] = 4
    if first:
        s["scores"].zero_()
        s["y"].zero_()
        s["hyp_len"].zero_()
    return {k: v.to(dev) for k, v in s.items()}


@pytest.mark.parametrize("V,K,Hd,first,ties", [(100, 4, 64, False, 0), (5000, 16, 128, False, 0), (333, 8, 64, True, 0),
                                                (5000, 16, 64, False, 3), (5000, 16, 64, False, 1), (700, 8, 64, False, 40)])
@pytest.mark.parametrize("terms", [3, 1, 4])
def test_fc2_logits_advance_equals_the_materialised_logits_advance(hip_device, V, K, Hd, first, terms, ties):
    """fc2 with row statistics + scaled logits (pika_dfc2_logits) + pika_beam_advance_logits (the thresholded row pass) ==
    materialised logits -> pika_beam_advance: same parents, symbols, finished lists; scores to fp32 rounding of the
    log-sum-exp.  ties > 0: logits take only that many distinct values
    (more candidates at the bound than the selection pool holds: the bound is raised / ties go by lowest column)."""
    from pika_amd.decoder.fused_step import PackedWeight
    from pika_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(V + K + terms)
    B, L, blk, S_steps = 3, 12, 0, 40
    R = B * K
    h = torch.randn(R, Hd, generator=g).to(hip_device)
    W = (torch.randn(V, Hd, generator=g) * 0.5).to(hip_device)
    bias = torch.randn(V, generator=g).to(hip_device)
    if ties:
        W.zero_()
        bias = torch.randint(0, ties, (V,), generator=g).float().to(hip_device)
    pw = PackedWeight(W, terms)
    sm = 0.8
    splits = lib.pika_dfc2_splits(V)
    ldl = splits * lib.pika_dfc2_cols_per_split()
    pmax2 = torch.empty(R * splits, device=hip_device)
    psum2 = torch.empty(R * splits, device=hip_device)
    slog = torch.full((R, ldl), 7.0, device=hip_device)
    _lib.check(lib.pika_dfc2_logits(h.data_ptr(), Hd, pw.buf.data_ptr(), bias.data_ptr(), R, V, Hd, terms, sm,
                                    pmax2.data_ptr(), psum2.data_ptr(), slog.data_ptr(), ldl, _st()), "pika_dfc2_logits")
    # the logits the same arithmetic produces (terms-term operands): via a plain dgemm, then the old advance
    from pika_amd.decoder.fused_step import DGemm
    logits = torch.empty(R, V, device=hip_device)
    d = DGemm()
    d.A, d.lda, d.W, d.bias, d.C, d.ldc = h.data_ptr(), Hd, pw.buf.data_ptr(), bias.data_ptr(), logits.data_ptr(), V
    d.M, d.N, d.K, d.terms, d.flags = R, V, Hd, terms, 0
    _lib.check(lib.pika_dgemm(ctypes.byref(d), _st()), "pika_dgemm")
    if terms == 3:
        want = h.double() @ W.double().t() + bias.double()
        assert (logits.double() - want).abs().max().item() <= 3e-6 * want.abs().max().item()
    # partial statistics
    x = (sm * logits).double()
    lse = torch.logsumexp(x, dim=1)
    pm = pmax2.view(R, splits).double()
    got_lse = (psum2.view(R, splits).double() * torch.exp(pm - pm.max(1, keepdim=True).values)).sum(1).log() + pm.max(1).values
    assert (got_lse - lse).abs().max().item() < 1e-5
    # (the plain product sums its reduction in another order than the vocabulary product)
    assert torch.allclose(slog[:, :V], sm * logits, rtol=0, atol=1e-5 * float(logits.abs().max()) + (2e-2 if terms == 1 else 0))
    assert bool(torch.all(slog[:, V:] == -float("inf")))

    def run(use_partials):
        st = _beam_inputs(B, K, V, L, torch.Generator().manual_seed(9), hip_device, first)
        ks_hist = torch.zeros(S_steps, B, K, dtype=torch.long, device=hip_device)
        ys_hist = torch.zeros(S_steps + 1, B, K, dtype=torch.long, device=hip_device)
        step_t = torch.full((1,), 0 if first else 3, dtype=torch.long, device=hip_device)
        fin_cap = K * S_steps + 1
        fin_score = torch.zeros(B, fin_cap, device=hip_device)
        fin_step = torch.zeros(B, fin_cap, dtype=torch.long, device=hip_device)
        fin_k = torch.zeros(B, fin_cap, dtype=torch.long, device=hip_device)
        fin_n = torch.zeros(B, dtype=torch.long, device=hip_device)
        prev_k = torch.zeros(B, K, dtype=torch.long, device=hip_device)
        y_raw = torch.zeros(B, K, dtype=torch.long, device=hip_device)
        stop = torch.zeros(1, dtype=torch.int32, device=hip_device)
        sync = torch.zeros(8, dtype=torch.int32, device=hip_device)
        max_hyp = torch.zeros(1, dtype=torch.long, device=hip_device)
        common = [st["scores"].data_ptr(), st["lm_scores"].data_ptr(), 1.0, st["y"].data_ptr(), st["t_idx"].data_ptr(),
                  st["num_frames"].data_ptr(), st["max_len"].data_ptr(), st["hyp"].data_ptr(), st["hyp_len"].data_ptr(), L,
                  ks_hist.data_ptr(), ys_hist.data_ptr(), step_t.data_ptr(), st["eos"].data_ptr(), fin_score.data_ptr(),
                  fin_step.data_ptr(), fin_k.data_ptr(), fin_n.data_ptr(), fin_cap, prev_k.data_ptr(), y_raw.data_ptr()]
        if use_partials == 2:
            _lib.check(lib.pika_beam_advance_logits(pmax2.data_ptr(), psum2.data_ptr(), slog.data_ptr(), ldl, splits, *common,
                                                    B, K, V, blk, 1, K, stop.data_ptr(), max_hyp.data_ptr(),
                                                    sync.data_ptr(), _st()), "pika_beam_advance_logits")
            assert int(step_t) == (1 if first else 4) and int(max_hyp) == int(st["hyp_len"].max())
        else:
            cand = torch.empty(B * K * K * 8, dtype=torch.uint8, device=hip_device)
            _lib.check(lib.pika_beam_advance(logits.data_ptr(), sm, int(first), *common, cand.data_ptr(), B, K, V, blk, 1,
                                             _st()), "pika_beam_advance")
        torch.cuda.synchronize()
        return dict(st, prev_k=prev_k, y_raw=y_raw, fin_n=fin_n, fin_score=fin_score, fin_k=fin_k, fin_step=fin_step)
    b, c = run(False), run(2)
    for k in ("prev_k", "y_raw", "y", "t_idx", "hyp", "hyp_len", "fin_n", "fin_k", "fin_step", "eos"):
        assert torch.equal(c[k], b[k]), k
    assert torch.allclose(c["scores"], b["scores"], rtol=0, atol=2e-5)
    assert torch.allclose(c["fin_score"], b["fin_score"], rtol=0, atol=2e-5)
    # the library's own statement of the launch's LDS need (what fused_step.supported() asks)
    assert lib.pika_beam_advance_logits_lds(K, L, splits) > 0 and lib.pika_beam_advance_logits_lds(65, L, splits) == 0


@pytest.mark.parametrize("pred_net", ["transformer", "rnn"])
@pytest.mark.parametrize("dec_terms", ["fp32", "bf16"])
def test_fused_search_equals_stepwise_search(hip_device, dec_terms, pred_net):
    """The launch-chain search (hipGraph, several steps per host read) and the op-by-op search return the same
    n-best lists for the tiny golden model, with and without graph replay -- conv-transformer prediction net (22
    launches per step) and the LSTM prediction net of the shipped recipes (4 + 2 per layer)."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from types import SimpleNamespace
    import decode_common as D
    from test_decode import build
    from decoder.transducer_decoder import TransducerDecoder
    from decoder.beam_transducer import GlobalScorer
    net = build(pred_net, hip_device)
    want_launches = 16 if pred_net == "transformer" else 4 + 2 * net.decoder.num_layers
    x, x_len = D.inputs()
    x, x_len_d = x.to(hip_device), x_len.to(hip_device)
    args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=0.0)
    for name in ("beam4", "beam8", "beam3_short"):
        cfg = D.SCENARIOS[name]
        outs = []
        for fused, graph in ((True, True), (True, False), (False, True)):
            d = TransducerDecoder(net, batch_size=x.shape[0], beam_size=cfg["beam"], n_best=cfg["n_best"], blk=0,
                                  global_scorer=GlobalScorer(), sm_scale=cfg["sm_scale"], cuda=True, beam_prune=True, args=args)
            d.fused_search, d.use_graph, d.decode_precision = fused, graph, dec_terms
            ret, _ = d.decode_batch(x, x_len_d, D.max_len(cfg, x_len))
            outs.append(D.pack(ret["predictions"], ret["scores"]))
            if fused:
                assert d.timing["launches_per_step"] == want_launches and d.timing["steps"] > 5
        if dec_terms == "fp32":
            for o in outs[1:]:
                assert np.array_equal(o["hyps"], outs[0]["hyps"]) and np.array_equal(o["lens"], outs[0]["lens"]), name
                assert np.allclose(o["scores"], outs[0]["scores"], atol=1e-4)
        else:       # bf16 operands: graph and eager runs of the SAME chain agree exactly; the op-by-op search rounds
            assert np.array_equal(outs[0]["hyps"], outs[1]["hyps"])          # elsewhere and may differ on near-ties


def test_backtrack_launch_gives_the_lists_of_the_host_walk(hip_device):
    """BeamState.results on the device (pika_beam_backtrack: one thread per n-best entry walks its back-pointers) against
    the host walk over copied histories, on random histories with ragged finishing steps and fill-up entries."""
    from pika_amd.decoder.beam_search import BeamState
    B, K, S, nb, V = 7, 16, 61, 16, 500
    g = torch.Generator().manual_seed(11)
    states = []
    for dev in (torch.device("cpu"), hip_device):
        bs = BeamState(B, K, 0, nb, [S + 5] * B, V, dev)
        states.append(bs)
    ks = torch.randint(0, K, (S, B, K), generator=g)
    ys = torch.randint(1, V, (S + 1, B, K), generator=g)
    ys[torch.rand(S + 1, B, K, generator=g) < 0.8] = 0
    fin_n = torch.randint(0, 30, (B,), generator=g)          # fewer than n_best: fill-ups; more: the best n_best
    fin_score = torch.randn(B, 40, generator=g)
    fin_step = torch.randint(1, S + 1, (B, 40), generator=g)
    fin_k = torch.randint(0, K, (B, 40), generator=g)
    scores = torch.randn(B, K, generator=g)
    for bs in states:
        bs.steps = S
        bs.ks_hist[:S] = ks.to(bs.device)
        bs.ys_hist[:S + 1] = ys.to(bs.device)
        bs.fin_n[:] = fin_n.to(bs.device)
        bs.fin_score[:, :40] = fin_score.to(bs.device)
        bs.fin_step[:, :40] = fin_step.to(bs.device)
        bs.fin_k[:, :40] = fin_k.to(bs.device)
        bs.scores[:] = scores.to(bs.device)
    (p0, s0), (p1, s1) = states[0].results(), states[1].results()
    assert [[[int(e) for e in h] for h in row] for row in p0] == [[[int(e) for e in h] for h in row] for row in p1]
    assert [[float(v) for v in row] for row in s0] == [[float(v) for v in row] for row in s1]
    assert all(hasattr(e, "item") for e in p1[0][0])
