"""Generates tests/golden/decode_fst.npz: the REFERENCE decoder + the REFERENCE SortedMatcher
(decoder/sorted_matcher.py, whose search/back-off logic is in-tree) run against a duck-typed FST
object exposing the OpenFST calls it makes (`arcs(state)` -> iterator with seek/done/value,
`final(state).value`) over the synthetic bigram of tests/fst_common.py.  `kaldi.fstext` itself is
only imported, never used, by the matcher: a stub module stands in.
    python tests/golden/make_fst_golden.py
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
k = types.ModuleType("kaldi"); k.fstext = types.ModuleType("kaldi.fstext")
sys.modules["kaldi"], sys.modules["kaldi.fstext"] = k, k.fstext
from oracle import pika_ref  # noqa: E402
import model_common as C  # noqa: E402
import decode_common as D  # noqa: E402
import fst_common as FC  # noqa: E402


DuckFst = FC.DuckFst


transducer, encoder, tdec, beam_mod, sm = pika_ref.load_reference(
    "trainer.model.transducer", "trainer.model.rnnt_tdnn_transformer",
    "decoder.transducer_decoder", "decoder.beam_transducer", "decoder.sorted_matcher")
n_states, arcs, finals, params = FC.bigram_arcs(C.V)
duck = DuckFst(n_states, arcs, finals)
matcher = sm.SortedMatcher(duck, **params)
out = {}
# matcher queries (row 15 proper)
q = []
rng = np.random.default_rng(0)
for _ in range(300):
    st, il = int(rng.integers(0, n_states)), int(rng.integers(1, C.V + 4))
    sc, ns = matcher.get_scores(st, il)
    fs, fstates = matcher.final_score(st)
    q.append((st, il, sc, ns, fs, fstates))
out["queries"] = np.array([(a, b) for a, b, *_ in q])
out["q_scores"] = np.array([np.pad(np.array(c, float), (0, 8 - len(c)), constant_values=np.nan) for _, _, c, *_ in q])
out["q_states"] = np.array([np.pad(np.array(d, int), (0, 8 - len(d)), constant_values=-1) for _, _, _, d, *_ in q])
out["q_final"] = np.array([np.pad(np.array(e, float), (0, 4 - len(e)), constant_values=np.nan) for *_, e, _ in q])
for dec in ("rnn", "transformer"):
    net = C.build(transducer, encoder, dec)
    net.load_state_dict(pika_ref.seeded_state_dict(net, C.SEED))
    D.tweak(net)
    net.eval()
    x, x_len = D.inputs()
    for name, (gs, lam, reward) in {"fused": (True, 0.5, 0.0), "fused_noscorer": (False, 0.3, 0.2)}.items():
        args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=reward)
        d = tdec.TransducerDecoder(net, batch_size=4, beam_size=4, n_best=4, blk=0,
                                   global_scorer=beam_mod.GlobalScorer() if gs else None, sm_scale=0.8,
                                   lm_scorer=matcher, lm_scorer_scale=lam, cuda=False, beam_prune=True, args=args)
        with torch.no_grad():
            ret, _ = d.decode_batch(x, x_len, [int(v) + 100 for v in x_len])
        for kk, v in D.pack(ret["predictions"], ret["scores"]).items():
            out["%s/%s/%s" % (dec, name, kk)] = v
        print(dec, name, [int(e) for e in ret["predictions"][0][0]][:20], float(ret["scores"][0][0]))
np.savez_compressed(os.path.join(HERE, "decode_fst.npz"), **out)
print("wrote decode_fst.npz")
