"""FST shallow fusion: our CSR matcher and fused beam search vs golden recorded
from the REFERENCE SortedMatcher + decoder on a duck-typed FST (tests/golden/make_fst_golden.py)."""
import os
import struct
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import model_common as C  # noqa: E402
import decode_common as D  # noqa: E402
import fst_common as FC  # noqa: E402
from test_decode import build  # noqa: E402

GOLD = os.path.join(HERE, "golden", "decode_fst.npz")


def matcher():
    from pika_amd.decoder.ngram_fst import NgramFst, SortedMatcher
    n, arcs, finals, params = FC.bigram_arcs(C.V)
    return SortedMatcher(NgramFst.from_arcs(n, arcs, finals), **params)


def test_matcher_queries_match_reference():
    z = np.load(GOLD)
    m = matcher()
    for (st, il), sc, ns, fs in zip(z["queries"], z["q_scores"], z["q_states"], z["q_final"]):
        scores, states = m.get_scores(int(st), int(il))
        k = int((~np.isnan(sc)).sum())
        assert states == ns[:k].tolist() and np.allclose(scores, sc[:k], atol=1e-6)
        f, _ = m.final_score(int(st))
        kf = int((~np.isnan(fs)).sum())
        assert np.allclose(f, fs[:kf], atol=1e-6, equal_nan=True)


def test_fst_file_formats(tmp_path):
    from pika_amd.decoder.ngram_fst import NgramFst
    n, arcs, finals, _ = FC.bigram_arcs(12, seed=4, n_succ=3)
    ref = NgramFst.from_arcs(n, arcs, finals)
    txt = tmp_path / "g.txt"
    with open(txt, "w") as f:
        for s, i, w, d in arcs:
            f.write("%d %d %d %d %r\n" % (s, d, i, i, w))
        for s, w in finals.items():
            f.write("%d %r\n" % (s, w))
    a = NgramFst.read_text(str(txt))
    # OpenFST binary (vector/standard, no symbol tables)
    b = tmp_path / "g.fst"
    with open(b, "wb") as f:
        def s_(t): return struct.pack("<i", len(t)) + t.encode()
        f.write(struct.pack("<i", 2125659606) + s_("vector") + s_("standard") + struct.pack("<iiQqqq", 2, 0, 0, 0, n, len(arcs)))
        for st in range(n):
            lo, hi = ref.offsets[st], ref.offsets[st + 1]
            f.write(struct.pack("<fq", float(ref.final[st]), hi - lo))
            for j in range(lo, hi):
                f.write(struct.pack("<iifi", int(ref.ilabel[j]), int(ref.ilabel[j]), float(ref.weight[j]), int(ref.nextstate[j])))
    c = NgramFst.read_binary(str(b))
    # ... and the same FST with EMBEDDED symbol tables (header flags 1 | 2; `fstcompile --keep_isymbols --keep_osymbols`):
    # OpenFST 1.6/1.7 SymbolTableImpl::Write -- int32 magic, name, int64 available key, int64 size, size x (symbol, int64 key)
    syms = ["<eps>"] + ["w%d" % i for i in range(1, 14)]

    def table(name):
        out = struct.pack("<i", 2125658996) + s_(name) + struct.pack("<qq", len(syms), len(syms))
        for k, sym in enumerate(syms):
            out += s_(sym) + struct.pack("<q", k)
        return out
    b2 = tmp_path / "g_syms.fst"
    with open(b2, "wb") as f:
        f.write(struct.pack("<i", 2125659606) + s_("vector") + s_("standard") + struct.pack("<iiQqqq", 2, 3, 0, 0, n, len(arcs)))
        f.write(table("words.txt") + table("words.txt"))
        for st in range(n):
            lo, hi = ref.offsets[st], ref.offsets[st + 1]
            f.write(struct.pack("<fq", float(ref.final[st]), hi - lo))
            for j in range(lo, hi):
                f.write(struct.pack("<iifi", int(ref.ilabel[j]), int(ref.ilabel[j]), float(ref.weight[j]), int(ref.nextstate[j])))
    d = NgramFst.read_binary(str(b2))
    assert d.isymbols[0] == "<eps>" and d.osymbols[13] == "w13" and len(d.isymbols) == len(syms) and c.isymbols is None
    for g in (a, c, d):
        assert np.array_equal(g.offsets, ref.offsets) and np.array_equal(g.ilabel, ref.ilabel)
        assert np.allclose(g.weight, ref.weight) and np.array_equal(g.nextstate, ref.nextstate)
        assert np.allclose(g.final, ref.final)


def run(dec, device):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from decoder.transducer_decoder import TransducerDecoder
    from decoder.beam_transducer import GlobalScorer
    z = np.load(GOLD)
    net = build(dec, device)
    x, x_len = D.inputs()
    for name, (gs, lam, reward) in {"fused": (True, 0.5, 0.0), "fused_noscorer": (False, 0.3, 0.2)}.items():
        args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=reward)
        d = TransducerDecoder(net, batch_size=4, beam_size=4, n_best=4, blk=0,
                              global_scorer=GlobalScorer() if gs else None, sm_scale=0.8, lm_scorer=matcher(),
                              lm_scorer_scale=lam, cuda=(device != "cpu"), beam_prune=True, args=args)
        ret, _ = d.decode_batch(x.to(device), x_len.to(device), [int(v) + 100 for v in x_len])
        got = D.pack(ret["predictions"], ret["scores"])
        pre = "%s/%s/" % (dec, name)
        assert np.array_equal(got["hyps"], z[pre + "hyps"]), (dec, name)
        assert np.allclose(got["scores"], z[pre + "scores"], rtol=1e-5, atol=2e-4), (dec, name)
        if device != "cpu":
            # the run above used the device-resident FST (one HIP launch per step, hipGraph-captured search);
            # the host-side state sets (the reference's own data structure) must give the same n-best lists
            assert d.timing["graphs"] > 0 or d.timing["steps"] <= 3
            d.fused_step = False
            ret2, _ = d.decode_batch(x.to(device), x_len.to(device), [int(v) + 100 for v in x_len])
            got2 = D.pack(ret2["predictions"], ret2["scores"])
            assert np.array_equal(got2["hyps"], got["hyps"]) and np.allclose(got2["scores"], got["scores"], atol=1e-5)


@pytest.mark.parametrize("dec", ["rnn", "transformer"])
def test_cpu_fused_decode_matches_reference(dec):
    run(dec, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "default"])
def test_gpu_fused_decode_matches_reference(hip_device, precision):
    """The tiny-model FST goldens in the exact mode and in the package default (the decoder switches to its own decode
    arithmetic either way; the full-width FST-fused golden: tests/test_decode_full.py)."""
    from pika_amd import gemm as G
    old = G.PRECISION
    if precision != "default":
        G.PRECISION = precision
    try:
        run("transformer", hip_device)
        run("rnn", hip_device)
    finally:
        G.PRECISION = old


@pytest.mark.gpu
def test_device_fst_advance_equals_host_state_sets(hip_device):
    """pika_fst_advance vs BeamState._fst_update/_fst_final (the restatement of beam_transducer.py:135-181 that is
    pinned by the golden above) on random parents / symbols over several steps: identical sets (same order),
    lm scores, finished scores."""
    from pika_amd.decoder.beam_search import BeamState, EOS
    B, K, V = 5, 6, C.V
    g = torch.Generator().manual_seed(17)
    dev_beam = BeamState(B, K, 0, 4, [50] * B, V, hip_device, lm_scorer=matcher(), nonblk_reward=0.2)
    host_beam = BeamState(B, K, 0, 4, [50] * B, V, torch.device("cpu"), lm_scorer=matcher(), nonblk_reward=0.2)
    assert dev_beam.fst_dev is not None and host_beam.fst_dev is None
    d = dev_beam.fst_dev
    for step in range(12):
        prev_k = torch.randint(0, K, (B, K), generator=g)
        y = torch.randint(0, V, (B, K), generator=g)
        y[torch.rand(B, K, generator=g) < 0.3] = 0                       # blanks keep the parent's set
        fin = torch.rand(B, K, generator=g) < 0.15
        scores = torch.randn(B, K, generator=g)
        # host
        host_beam._fst_update(prev_k, y)
        want_fin = scores + 0.5 * host_beam._fst_final(fin)
        # device: emulate what pika_beam_advance leaves behind (y with eos, finished entries appended in slot order)
        dev_beam.scores.copy_(scores)
        d["y_raw"].copy_(y)
        dev_beam.y.copy_(torch.where(fin, torch.full_like(y, EOS), y))
        n_old = dev_beam.fin_n.clone()
        dev_beam.fin_n += fin.sum(1).to(hip_device)
        dev_beam._fst_advance_device(prev_k.to(hip_device), 0.5)
        assert torch.allclose(dev_beam.lm_scores.cpu(), host_beam.lm_scores, atol=0, rtol=0)
        assert torch.allclose(dev_beam.scores.cpu(), torch.where(fin, want_fin, scores), atol=1e-6)
        sn, st, cs = d["set_n"].cpu().view(B, K), d["set_st"].cpu().view(B, K, -1), d["set_cs"].cpu().view(B, K, -1)
        for b in range(B):
            pos = int(n_old[b])
            for i in range(K):
                hs = host_beam.state_sets[b][i]
                assert int(sn[b, i]) == len(hs)
                assert st[b, i, :len(hs)].tolist() == list(hs.keys())
                assert cs[b, i, :len(hs)].tolist() == list(hs.values())
                if fin[b, i]:
                    assert abs(float(dev_beam.fin_score[b, pos]) - float(want_fin[b, i])) < 1e-6
                    pos += 1
    assert int(d["err"].item()) == 0


@pytest.mark.gpu
def test_device_fst_overflow_is_flagged(hip_device):
    """A back-off chain 11 levels deep yields 11 next states for one symbol: more than the device arrays hold.
    The kernel must raise its flag (TransducerDecoder then re-decodes the batch on the host path)."""
    from pika_amd.decoder.beam_search import BeamState
    from pika_amd.decoder.ngram_fst import NgramFst, SortedMatcher
    V, lab, bo = 30, 5, 40
    arcs, finals = [], {}
    for i in range(11):
        arcs.append((i, lab + 1, 1.0 + i, 20 + i))
        if i < 10:
            arcs.append((i, bo, 0.5, i + 1))
    for i in range(20, 31):
        finals[i] = 0.0
    m = SortedMatcher(NgramFst.from_arcs(31, arcs, finals), max_num_arcs=V + 4, max_id=bo + 1, backoff_id=bo,
                      disambig_ids=[])
    beam = BeamState(1, 2, 0, 1, [50], V, hip_device, lm_scorer=m)
    assert beam.fst_dev is not None and not beam.fst_overflowed()
    beam.fst_dev["y_raw"].fill_(lab)
    beam.y.fill_(lab)
    beam._fst_advance_device(torch.zeros(1, 2, dtype=torch.long, device=hip_device), 0.5)
    assert beam.fst_overflowed()
    host = BeamState(1, 2, 0, 1, [50], V, torch.device("cpu"), lm_scorer=m)
    host._fst_update(torch.zeros(1, 2, dtype=torch.long), torch.full((1, 2), lab))
    assert len(host.state_sets[0][0]) == 11
