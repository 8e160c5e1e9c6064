"""Per-token kernels of LAS rescoring (include/pika_las.h) vs the torch formulas of the reference modules
(modules/global_attention.py:162-248 "mlp" attention; nn.LSTMCell as stacked by modules/stacked_rnn.py:20-34), and
the fused scoring pass of pika_amd.model.las against its own op-by-op pass on a ragged decode batch."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import las_common as LC  # noqa: E402
from oracle.pika_ref import seeded_state_dict  # noqa: E402


@pytest.mark.parametrize("N,H", [(1, 4), (5, 32), (1024, 1024), (7, 100)])
def test_lstm_cell(hip_device, N, H):
    from pika_amd import _lib
    g = torch.Generator().manual_seed(N + H)
    gates = (torch.randn(N, 4 * H + 8, generator=g) * 3).to(hip_device)          # pitch > 4H
    gates[0, :4] = torch.tensor([100.0, -100.0, 50.0, -50.0])                    # saturated gates stay finite
    c_prev = torch.randn(N, H, generator=g).to(hip_device)
    i, f, gg, o = gates[:, :4 * H].double().chunk(4, dim=1)
    c_ref = torch.sigmoid(f) * c_prev.double() + torch.sigmoid(i) * torch.tanh(gg)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    c = c_prev.clone()
    h1 = torch.full((N, H + 4), -7.0, device=hip_device)
    h2 = torch.full((N, 2 * H), -7.0, device=hip_device)
    with torch.cuda.device(hip_device):
        _lib.check(_lib.lib().pika_lstm_cell(gates.data_ptr(), gates.stride(0), c.data_ptr(), c.data_ptr(), h1.data_ptr(),
                                             h1.stride(0), h2[:, H:].data_ptr(), h2.stride(0), N, H, None, None, None,
                                             torch.cuda.current_stream().cuda_stream), "pika_lstm_cell")
    assert (c.double() - c_ref).abs().max() < 2e-6 * max(1.0, c_ref.abs().max().item())
    assert (h1[:, :H].double() - h_ref).abs().max() < 2e-6 and torch.equal(h1[:, :H], h2[:, H:])
    assert bool((h1[:, H:] == -7.0).all()) and bool((h2[:, :H] == -7.0).all())


@pytest.mark.parametrize("B,S,D,nper", [(1, 5, 4, [1]), (3, 23, 32, [3, 2, 1]), (4, 240, 1024, [16, 16, 5, 16]),
                                        (2, 300, 100, [7, 9]),
                                        (5, 40, 64, [300, 301, 299, 150, 3])])     # > 1024 queries: four per workgroup
def test_mlp_attention(hip_device, B, S, D, nper):
    """Queries of several utterances, ragged source lengths, against the formula in float64 (las.py:GlobalAttention "mlp")."""
    from pika_amd import _lib
    g = torch.Generator().manual_seed(B * 100 + S + D)
    owner = torch.tensor([b for b in range(B) for _ in range(nper[b])], dtype=torch.int32)
    N = owner.numel()
    lens = torch.tensor([max(1, S - 7 * b) for b in range(B)], dtype=torch.int32)
    wq = torch.randn(N, D, generator=g)
    proj = torch.randn(B, S, D, generator=g)
    ctx = torch.randn(B, S, D, generator=g)
    v = torch.randn(D, generator=g) * 0.3
    align = (torch.tanh(wq.double().unsqueeze(1) + proj.double()[owner.long()]) * v.double()).sum(-1)      # (N,S)
    mask = torch.arange(S).unsqueeze(0) < lens[owner.long()].unsqueeze(1)
    a_ref = torch.softmax(align.masked_fill(~mask, float("-inf")), -1)
    c_ref = torch.bmm(a_ref.unsqueeze(1), ctx.double()[owner.long()]).squeeze(1)
    dev = [t.to(hip_device) for t in (wq, proj, ctx, owner, lens, v)]
    # utterance by utterance (pika_las_mlp_attention_by_utterance): the list ordered by utterance (a subset: utterance 0
    # loses a query, the last utterance all of them), behind an offset, with the step's table of utterance ranges
    lib = _lib.lib()
    keep = [i for i in range(N) if not (owner[i] == B - 1 and B > 1)]
    if nper[0] > 1:
        keep.remove(0)
    off, step_t = 3, 2
    qlist = torch.cat([torch.full((off,), -1, dtype=torch.int32), torch.tensor(keep, dtype=torch.int32)])
    own_list = owner[keep].numpy()
    uoff = np.zeros((step_t + 1, B + 1), np.int32)
    uoff[step_t] = np.searchsorted(own_list, np.arange(B + 1))
    out3 = torch.full((N, 2 * D), -7.0, device=hip_device)
    with torch.cuda.device(hip_device):
        work = torch.empty(int(lib.pika_las_attention_work_floats(N + 4, S, D)), device=hip_device)
        ql_d, uo_d = qlist.to(hip_device), torch.from_numpy(uoff).to(hip_device)
        n_d = torch.tensor([len(keep)], dtype=torch.int32, device=hip_device)
        off_d = torch.tensor([off], dtype=torch.int32, device=hip_device)
        st_d = torch.tensor([step_t], dtype=torch.int32, device=hip_device)
        _lib.check(lib.pika_las_mlp_attention_by_utterance(
            dev[0].data_ptr(), D, dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), ql_d.data_ptr(),
            uo_d.data_ptr(), dev[5].data_ptr(), out3.data_ptr(), 2 * D, work.data_ptr(), N + 4, B, S, D, n_d.data_ptr(),
            off_d.data_ptr(), st_d.data_ptr(), torch.cuda.current_stream().cuda_stream), "pika_las_mlp_attention_by_utterance")
    assert (out3[keep][:, :D].double().cpu() - c_ref[keep]).abs().max() < 1e-5
    assert bool((out3[:, D:] == -7.0).all())
    rest = torch.ones(N, dtype=torch.bool)
    rest[keep] = False
    assert bool((out3[rest] == -7.0).all())


def test_fused_scoring_pass_equals_the_op_by_op_pass(hip_device, monkeypatch):
    """score_nbest_batch on a ragged decode batch: the per-token kernel chain against the op-by-op decoder loop of the
    same module (PIKA_LAS_FUSED=0), exact arithmetic mode; and score_nbest (one utterance) the same way."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from trainer.model import las
    from pika_amd import gemm as G
    old, G.PRECISION = G.PRECISION, "fp32"
    try:
        net = las.Net(LC.opt("mlp"), LC.C_IN, LC.V, LC.PAD)
        net.load_state_dict(seeded_state_dict(net, 31, scale=0.3))
        net = net.to(hip_device).eval()
        g = torch.Generator().manual_seed(5)
        lens = [17, 23, 9]
        src = torch.zeros(23, 3, LC.C_IN)
        for b, n in enumerate(lens):
            src[:n, b] = torch.randn(n, LC.C_IN, generator=g)
        src = src.to(hip_device)
        hyps = [[[3, 7, 7, 12], [5], []], [[8, 1, 30, 2, 2, 19, 4], [2, 2]], [[11]]]
        res = {}
        for fused in ("1", "0"):
            monkeypatch.setenv("PIKA_LAS_FUSED", fused)
            res[fused] = (net.score_nbest_batch(src, lens, hyps, LC.SOS, LC.EOS),
                          net.score_nbest(src[:17, 0:1], hyps[0], LC.SOS, LC.EOS))
        for a, b in zip(res["1"][0], res["0"][0]):
            for x, y in zip(a, b):
                assert np.allclose(x, y, rtol=2e-5, atol=2e-5), (x, y)
        for x, y in zip(res["1"][1], res["0"][1]):
            assert np.allclose(x, y, rtol=2e-5, atol=2e-5)
    finally:
        G.PRECISION = old


@pytest.mark.parametrize("S,B,D,H,In,layers", [(9, 5, 2, 128, 64, 1), (40, 37, 2, 512, 256, 2), (25, 16, 1, 256, 128, 2),
                                               (12, 64, 2, 384, 64, 1),
                                               (14, 150, 2, 512, 128, 2)])     # > 64 rows: three row blocks of the kernel
def test_blstm_encoder_matches_nn_lstm(hip_device, monkeypatch, S, B, D, H, In, layers):
    """The persistent-kernel encoder (pika_blstm_layer: recurrent weights in registers, hidden states exchanged between
    workgroups inside the launch) against torch's nn.LSTM over the same packed sequences in float64 on the CPU:
    outputs (zeros at padded positions), final hidden and cell states of every layer and direction; ragged lengths,
    a batch that does not fill its last 16-row block, an utterance of length 1."""
    from pika_amd.model import las
    torch.manual_seed(S * 1000 + B)
    enc = las.LASRNNEncoder("LSTM", D == 2, layers, D * H, 0.0, In).eval()
    for p in enc.parameters():                                   # recurrent gains well above the default 1/sqrt(H)
        p.data.uniform_(-2.0 / H ** 0.5, 2.0 / H ** 0.5)
    x = torch.randn(S, B, In)
    lens = torch.randint(1, S + 1, (B,))
    lens[0], lens[B // 2] = S, 1
    ref = enc.double()
    with torch.no_grad():
        (h_ref, c_ref), out_ref = ref(x.double(), lens)
    enc = enc.float().to(hip_device)
    with torch.no_grad():
        assert enc._fused_ok(x.to(hip_device), lens, None)
        (h, c), out = enc(x.to(hip_device), lens)
    assert enc.status_ok() and not enc._no_fused
    assert out.shape == out_ref.shape and h.shape == h_ref.shape
    for got, want, name in ((out, out_ref, "out"), (h, h_ref, "h_n"), (c, c_ref, "c_n")):
        err = (got.double().cpu() - want).abs().max().item()
        assert err < 2e-5 * max(1.0, want.abs().max().item()), (name, err)
    pad = torch.arange(out.shape[0]).view(-1, 1) >= lens.view(1, -1)
    assert bool((out.cpu()[pad] == 0).all())
    # the library path (what PIKA_LAS_BLSTM=0 runs) agrees too
    monkeypatch.setenv("PIKA_LAS_BLSTM", "0")
    with torch.no_grad():
        assert not enc._fused_ok(x.to(hip_device), lens, None)
        (h2, c2), out2 = enc(x.to(hip_device), lens)
    assert (out2 - out).abs().max().item() < 5e-5 and (h2 - h).abs().max().item() < 5e-5


def test_prefix_sharing_gives_the_values_of_scoring_every_entry_from_scratch(hip_device, monkeypatch):
    """n-best entries that share prefixes share decoder rows (tries per utterance, pika_las_fork_rows at the step an entry
    leaves the shared prefix).  Lists with common prefixes of every kind -- duplicates, an entry that is a strict prefix of
    an earlier and of a later one, entries that diverge at the first / last token, an empty entry, an utterance without
    any sharing -- against the same pass with sharing off (every entry its own rows from step 0): identical values."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from trainer.model import las
    net = las.Net(LC.opt("mlp"), LC.C_IN, LC.V, LC.PAD)
    net.load_state_dict(seeded_state_dict(net, 32, scale=0.3))
    net = net.to(hip_device).eval()
    g = torch.Generator().manual_seed(6)
    lens = [17, 23, 9, 12]
    src = torch.zeros(23, 4, LC.C_IN)
    for b, n in enumerate(lens):
        src[:n, b] = torch.randn(n, LC.C_IN, generator=g)
    src = src.to(hip_device)
    base = [3, 7, 7, 12, 9, 4, 21, 5]
    hyps = [[base, base[:5] + [30, 1], base, base[:3], base + [2, 2], [8] + base[1:], base[:7] + [6], []],
            [[8, 1, 30, 2, 2, 19, 4], [8, 1, 30, 2, 2, 19], [8, 1, 30, 2, 2, 19, 4, 4, 4], [8, 1]],
            [[11], [12], [13, 11]],
            [[5, 5, 5, 5], [5, 5, 5, 5], [5, 5, 5], [5, 5, 5, 6]]]
    res = {}
    for share in ("1", "0"):
        monkeypatch.setenv("PIKA_LAS_SHARE_PREFIXES", share)
        res[share] = net.score_nbest_batch(src, lens, hyps, LC.SOS, LC.EOS)
        info = dict(net.last_pass)
        assert info["shared"] == (share == "1") and info["pairs"] == sum(len(h) + 1 for row in hyps for h in row)
        if share == "1":
            assert info["row_steps"] == 43          # the distinct prefixes of the lists above (114 pairs)
    for a, b in zip(res["1"], res["0"]):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert len(x) == len(y) and np.allclose(x, y, rtol=0, atol=1e-6), (x, y)
    # one utterance through score_nbest
    monkeypatch.setenv("PIKA_LAS_SHARE_PREFIXES", "1")
    one = net.score_nbest(src[:17, 0:1], hyps[0], LC.SOS, LC.EOS)
    for x, y in zip(one, res["0"][0]):
        assert np.allclose(x, y, rtol=0, atol=2e-5)


def test_fork_rows_and_row_lists(hip_device):
    """pika_las_fork_rows (segments of row dst <- row src for the forks of step t only) and the gather-list forms of
    pika_lstm_cell / pika_las_embed_rows (launch row e stands for row rowlist[off + e])."""
    from pika_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(11)
    N, W1, W2 = 13, 40, 24
    a = torch.randn(N, W1, generator=g).to(hip_device)
    b = torch.randn(N, W2, generator=g).to(hip_device)
    a0, b0 = a.clone(), b.clone()
    step = torch.tensor([2, 5, 0, 0], dtype=torch.int32, device=hip_device)        # (n = 0 would make the launch a no-op)
    fork_off = torch.tensor([0, 0, 1, 4, 4], dtype=torch.int32, device=hip_device)       # step 2: forks 1..3
    dst = torch.tensor([9, 3, 7, 11], dtype=torch.int32, device=hip_device)
    src = torch.tensor([0, 1, 1, 5], dtype=torch.int32, device=hip_device)
    st = torch.cuda.current_stream().cuda_stream
    base = (ctypes.c_void_p * 2)(a.data_ptr(), b.data_ptr())
    ld = (ctypes.c_longlong * 2)(W1, W2)
    col0 = (ctypes.c_int * 2)(8, 0)
    ncols = (ctypes.c_int * 2)(16, 24)
    with torch.cuda.device(hip_device):
        _lib.check(lib.pika_las_fork_rows(step.data_ptr(), fork_off.data_ptr(), dst.data_ptr(), src.data_ptr(), 3, 2,
                                          base, ld, col0, ncols, st), "pika_las_fork_rows")
    want_a, want_b = a0.clone(), b0.clone()
    for d_, s_ in ((3, 1), (7, 1), (11, 5)):
        want_a[d_, 8:24] = a0[s_, 8:24]
        want_b[d_] = b0[s_]
    assert torch.equal(a, want_a) and torch.equal(b, want_b)
    # LSTM cell on a gather list
    H = 8
    gates = torch.randn(N, 4 * H, generator=g).to(hip_device)
    c = torch.randn(N, H, generator=g).to(hip_device)
    c_ref = c.clone()
    h = torch.full((N, H), -7.0, device=hip_device)
    rl = torch.tensor([-1, -1, 4, 12, 0], dtype=torch.int32, device=hip_device)
    off = torch.tensor([2], dtype=torch.int32, device=hip_device)
    n = torch.tensor([3], dtype=torch.int32, device=hip_device)
    with torch.cuda.device(hip_device):
        _lib.check(lib.pika_lstm_cell(gates.data_ptr(), 4 * H, c.data_ptr(), c.data_ptr(), h.data_ptr(), H, None, 0, N, H,
                                      n.data_ptr(), rl.data_ptr(), off.data_ptr(), st), "pika_lstm_cell")
    i_, f_, g_, o_ = gates.double().chunk(4, dim=1)
    cn = torch.sigmoid(f_) * c_ref.double() + torch.sigmoid(i_) * torch.tanh(g_)
    hn = torch.sigmoid(o_) * torch.tanh(cn)
    rows = [4, 12, 0]
    others = [r for r in range(N) if r not in rows]
    assert (c[rows].double() - cn[rows]).abs().max() < 2e-6 and (h[rows].double() - hn[rows]).abs().max() < 2e-6
    assert torch.equal(c[others], c_ref[others]) and bool((h[others] == -7.0).all())
    # embedding rows on the step's list
    E, V = 12, 20
    emb = torch.randn(V, E, generator=g).to(hip_device)
    tokens = torch.randint(0, V, (4, N), generator=g).to(hip_device)
    x0 = torch.full((N, E + 4), -7.0, device=hip_device)
    crow = torch.full((N,), -1, dtype=torch.long, device=hip_device)
    step = torch.tensor([2, 3, 2, 0], dtype=torch.int32, device=hip_device)              # t = 2, n = 3, list offset 2
    with torch.cuda.device(hip_device):
        _lib.check(lib.pika_las_embed_rows(step.data_ptr(), tokens.data_ptr(), emb.data_ptr(), x0.data_ptr(), E + 4,
                                           crow.data_ptr(), N, E, rl.data_ptr(), st), "pika_las_embed_rows")
    for r in rows:
        assert torch.equal(x0[r, :E], emb[tokens[2, r]]) and int(crow[r]) == 2 * N + r
    assert bool((x0[others] == -7.0).all()) and bool((crow[others] == -1).all()) and bool((x0[:, E:] == -7.0).all())
