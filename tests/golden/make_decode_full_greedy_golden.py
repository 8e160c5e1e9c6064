"""Generates tests/golden/decode_full_greedy.npz: the REFERENCE decoder (decoder/transducer_decoder.py:66-186 with
beam_size = 1, n_best = 1 -- greedy search, the case north_star words as bit-exact) on the FULL-WIDTH model and inputs
of tests/decode_full_common.py, CPU fp32.  Also records, along the greedy path of every utterance, the smallest gap
between the best and the second-best log-probability of a step (a decision whose gap is below the fp32 noise of a
1024-term dot product would make "identical" a statement about rounding; the gaps are printed and stored).
    python tests/golden/make_decode_full_greedy_golden.py
"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pika_ref  # noqa: E402
import decode_common as D  # noqa: E402
import decode_full_common as F  # noqa: E402

transducer, tdec, beam_mod = pika_ref.load_reference("trainer.model.transducer", "decoder.transducer_decoder",
                                                     "decoder.beam_transducer")
net = F.build(transducer, pika_ref.seeded_state_dict)
x, x_len = F.inputs()
args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=0.0)
gaps = []
real_advance = beam_mod.BeamMergeTransducer.advance


def advance(self, word_probs, *a, **k):      # observe the decision margin of every step (no change of behaviour)
    top2 = torch.topk(word_probs.reshape(-1).float(), 2).values
    gaps.append(float(top2[0] - top2[1]))
    return real_advance(self, word_probs, *a, **k)


beam_mod.BeamMergeTransducer.advance = advance
d = tdec.TransducerDecoder(net, batch_size=F.B, beam_size=1, n_best=1, blk=0, global_scorer=beam_mod.GlobalScorer(),
                           sm_scale=F.SM_SCALE, cuda=False, beam_prune=True, args=args)
t0 = time.time()
with torch.no_grad():
    ret, enc = d.decode_batch(x, x_len, F.max_len(x_len))
print("reference greedy decode: %.1f s, %d decisions, smallest margin %.3g, 10 smallest %s" % (
    time.time() - t0, len(gaps), min(gaps), ["%.3g" % g for g in sorted(gaps)[:10]]))
out = D.pack(ret["predictions"], ret["scores"])
out["min_margin"] = np.array(min(gaps))
for b in range(F.B):
    h = [int(e) for e in ret["predictions"][b][0]]
    print(b, "greedy: %d symbols, %d labels, score %.4f" % (len(h), sum(1 for e in h if e), float(ret["scores"][b][0])))
np.savez_compressed(os.path.join(HERE, "decode_full_greedy.npz"), **out)
print("wrote decode_full_greedy.npz")
