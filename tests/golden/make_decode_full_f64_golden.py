"""Generates tests/golden/decode_full_f64.npz: the REFERENCE decoder (decoder/transducer_decoder.py +
decoder/beam_transducer.py from /root/reference under the shims of oracle/pika_ref.py) on the full-width model and the four
utterances of tests/decode_full_common.py with model.double() -- the same search in float64.  Recorded per utterance: the
WHOLE finished list (not only the n-best head), sorted by float64 score.

Why: the fp32 reference list (decode_full.npz) holds pairs of entries closer in score than fp32 arithmetic resolves (sums of
~150 log-probs of |logit| ~ 30).  Which order of such a pair is RIGHT is a question for float64; tests/test_decode_full.py
uses this file to judge the ranks the GPU returns where they differ from the fp32 reference's.
    python tests/golden/make_decode_full_f64_golden.py          (a few minutes of CPU)
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
import decode_full_common as F  # noqa: E402

transducer, tdec, beam_mod = pika_ref.load_reference("trainer.model.transducer", "decoder.transducer_decoder",
                                                     "decoder.beam_transducer")
torch.set_num_threads(8)
net = F.build(transducer, pika_ref.seeded_state_dict).double()
x, x_len = F.inputs()
args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=0.0)
d = tdec.TransducerDecoder(net, batch_size=F.B, beam_size=F.BEAM, n_best=F.BEAM, blk=0,
                           global_scorer=beam_mod.GlobalScorer(), sm_scale=F.SM_SCALE, cuda=False, beam_prune=True,
                           args=args)
beams = []
from_beam = d._from_beam
d._from_beam = lambda beam: (beams.extend(beam), from_beam(beam))[1]
t0 = time.time()
with torch.no_grad():
    ret, enc = d.decode_batch(x.double(), x_len, F.max_len(x_len))
print("reference decode in float64: %.1f s" % (time.time() - t0))
assert enc.dtype == torch.float64 and len(beams) == F.B
out = {}
nmax = max(len(b.finished) for b in beams)
lmax = max(len(b.next_ys) for b in beams)
hyps = np.full((F.B, nmax, lmax), -9, np.int64)
lens = np.zeros((F.B, nmax), np.int64)
scores = np.full((F.B, nmax), -np.inf, np.float64)
count = np.zeros(F.B, np.int64)
for i, b in enumerate(beams):
    sc, ks = b.sort_finished(minimum=F.BEAM)
    count[i] = len(ks)
    for j, (s, (t, k)) in enumerate(zip(sc, ks)):
        h = [int(e) for e in b.get_hyp(t, k)][:-1]              # strip the trailing eos, as _from_beam does
        hyps[i, j, :len(h)] = h
        lens[i, j] = len(h)
        scores[i, j] = float(s)
    print(i, "finished entries %d; top-1 score %.6f, gap to #2 %.3g" % (count[i], scores[i, 0], scores[i, 0] - scores[i, 1]))
np.savez_compressed(os.path.join(HERE, "decode_full_f64.npz"), hyps=hyps, lens=lens, scores=scores, count=count,
                    enc_sample=enc[:, ::7, ::37].numpy())
print("wrote decode_full_f64.npz")
