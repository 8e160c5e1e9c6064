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
    for g in (a, c):
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


@pytest.mark.parametrize("dec", ["rnn", "transformer"])
def test_cpu_fused_decode_matches_reference(dec):
    run(dec, "cpu")


@pytest.mark.gpu
def test_gpu_fused_decode_matches_reference(hip_device):
    from pika_amd import gemm as G
    old, G.PRECISION = G.PRECISION, "fp32"
    try:
        run("transformer", hip_device)
    finally:
        G.PRECISION = old
