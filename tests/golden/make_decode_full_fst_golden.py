"""Generates tests/golden/decode_full_fst.npz: the REFERENCE decoder (decoder/transducer_decoder.py +
decoder/beam_transducer.py:136-159,167-177 FST update / final cost) with the REFERENCE SortedMatcher
(decoder/sorted_matcher.py) over a duck-typed FST holding the seeded back-off bigram of tests/fst_common.py over the
4999 labels, fused into the search of the FULL-WIDTH model of tests/decode_full_common.py (V = 5000, H = 1024, beam 16,
n-best 16, B = 4) -- the configuration bench.py's configs[4] leg runs, CPU fp32.  Takes several minutes.
    python tests/golden/make_decode_full_fst_golden.py
"""
import os
import sys
import time
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
import decode_common as D  # noqa: E402
import decode_full_common as F  # noqa: E402
import fst_common as FC  # noqa: E402

transducer, tdec, beam_mod, sm = pika_ref.load_reference("trainer.model.transducer", "decoder.transducer_decoder",
                                                         "decoder.beam_transducer", "decoder.sorted_matcher")
net = F.build(transducer, pika_ref.seeded_state_dict)
x, x_len = F.inputs()
n_states, arcs, finals, params = FC.bigram_arcs(F.V)
matcher = sm.SortedMatcher(FC.DuckFst(n_states, arcs, finals), **params)
args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=F.FST_REWARD)
d = tdec.TransducerDecoder(net, batch_size=F.B, beam_size=F.BEAM, n_best=F.BEAM, blk=0,
                           global_scorer=beam_mod.GlobalScorer(), sm_scale=F.SM_SCALE, lm_scorer=matcher,
                           lm_scorer_scale=F.FST_SCALE, cuda=False, beam_prune=True, args=args)
t0 = time.time()
with torch.no_grad():
    ret, enc = d.decode_batch(x, x_len, F.max_len(x_len))
print("reference FST-fused decode: %.1f s" % (time.time() - t0))
out = D.pack(ret["predictions"], ret["scores"])
plain = np.load(os.path.join(HERE, "decode_full.npz"))
for b in range(F.B):
    h = [int(e) for e in ret["predictions"][b][0]]
    L0 = int(plain["lens"][b, 0])
    same = len(h) == L0 and h == [int(e) for e in plain["hyps"][b, 0, :L0]]
    print(b, "top-1: %d symbols, %d labels, score %.4f; gap to #2 %.4g; %s the top-1 of the search without the FST" % (
        len(h), sum(1 for e in h if e), float(ret["scores"][b][0]),
        float(ret["scores"][b][0]) - float(ret["scores"][b][1]), "SAME as" if same else "differs from"))
np.savez_compressed(os.path.join(HERE, "decode_full_fst.npz"), **out)
print("wrote decode_full_fst.npz")
