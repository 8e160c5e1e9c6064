"""Generates tests/golden/las_full.npz: per-token log-probs of the REFERENCE LAS rescorer (trainer/model/las.py:51-90,
522-565, 600-683 through decoder/transducer_decoder.py:219-236 `las_rescore`, one hypothesis and one B = 1 encoder pass
at a time, as decode_transducer.py:136-156 calls it) at the WIDTH bench.py's configs[4] leg runs (tests/las_full_common.py):
a forward rescorer on the n-best entries and a backward rescorer on the reversed entries.  CPU fp32, ~1 minute.
    python tests/golden/make_las_full_golden.py
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
import las_full_common as LF  # noqa: E402

las, tdec = pika_ref.load_reference("trainer.model.las", "decoder.transducer_decoder")
fw = LF.build(las, pika_ref.seeded_state_dict, LF.SEED_FW)
bw = LF.build(las, pika_ref.seeded_state_dict, LF.SEED_BW)
args = SimpleNamespace(las_rescorer=fw, las_rescorer_bw=bw, bilas_rescorer=None)
d = tdec.TransducerDecoder(None, 1, 1, args=args)
src, lens, hyps = LF.inputs()
out = {}
t0 = time.time()
lo, hi = 0.0, -1e9
with torch.no_grad():
    for b, n in enumerate(lens):
        x = src[:n, b:b + 1]                                            # (T', 1, C): decode_transducer.py:139
        for j, h in enumerate(hyps[b]):
            tgt = torch.LongTensor([LF.SOS] + h + [LF.EOS]).unsqueeze(-1).unsqueeze(-1)
            out["fw/%d/%d" % (b, j)] = np.array(d.las_rescore(x, tgt), np.float64)
            tgt = torch.LongTensor([LF.SOS] + h[::-1] + [LF.EOS]).unsqueeze(-1).unsqueeze(-1)
            out["bw/%d/%d" % (b, j)] = np.array(d.las_rescore(x, tgt, bw=True), np.float64)
            for k in ("fw", "bw"):
                v = out["%s/%d/%d" % (k, b, j)]
                lo, hi = min(lo, v.min()), max(hi, v.max())
print("reference rescoring: %.1f s; token log-probs span [%.3f, %.3f]" % (time.time() - t0, lo, hi))
print("utterance 0, entry 0:", out["fw/0/0"][:6])
np.savez_compressed(os.path.join(HERE, "las_full.npz"), **out)
print("wrote las_full.npz")
