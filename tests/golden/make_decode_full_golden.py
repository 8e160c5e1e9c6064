"""Generates tests/golden/decode_full.npz by running the REFERENCE decoder (decoder/transducer_decoder.py +
decoder/beam_transducer.py, imported from /root/reference under the shims of oracle/pika_ref.py) on the FULL-WIDTH
model of tests/decode_full_common.py (V = 5000, H = 1024, beam 16, n-best 16, B = 4), CPU fp32.  Takes a few minutes.
    python tests/golden/make_decode_full_golden.py
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
d = tdec.TransducerDecoder(net, batch_size=F.B, beam_size=F.BEAM, n_best=F.BEAM, blk=0,
                           global_scorer=beam_mod.GlobalScorer(), sm_scale=F.SM_SCALE, cuda=False, beam_prune=True,
                           args=args)
t0 = time.time()
with torch.no_grad():
    ret, enc = d.decode_batch(x, x_len, F.max_len(x_len))
print("reference decode: %.1f s" % (time.time() - t0))
out = D.pack(ret["predictions"], ret["scores"])
out["enc_sample"] = enc[:, ::7, ::37].numpy()          # a thin slice of the encoder output (activation parity)
for b in range(F.B):
    h = [int(e) for e in ret["predictions"][b][0]]
    print(b, "top-1: %d symbols, %d labels, score %.4f; gap to #2 %.4g" % (
        len(h), sum(1 for e in h if e), float(ret["scores"][b][0]), float(ret["scores"][b][0]) - float(ret["scores"][b][1])))
np.savez_compressed(os.path.join(HERE, "decode_full.npz"), **out)
print("wrote decode_full.npz")
