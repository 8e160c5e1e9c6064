"""Generates tests/golden/decode_full_wide.npz: the REFERENCE decoder (decoder/transducer_decoder.py + beam_transducer.py,
imported from /root/reference under the shims of oracle/pika_ref.py) on the full-width model of tests/decode_full_common.py at
the BENCHMARKED batch: B = 64 utterances (1.4-1.9 s) x beam 16 = 1024 beam rows, CPU fp32.
    python tests/golden/make_decode_full_wide_golden.py
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
torch.set_num_threads(8)
net = F.build(transducer, pika_ref.seeded_state_dict)
x, x_len = F.inputs_wide()
args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=0.0)
d = tdec.TransducerDecoder(net, batch_size=F.WIDE_B, beam_size=F.BEAM, n_best=F.BEAM, blk=0,
                           global_scorer=beam_mod.GlobalScorer(), sm_scale=F.SM_SCALE, cuda=False, beam_prune=True,
                           args=args)
t0 = time.time()
with torch.no_grad():
    ret, enc = d.decode_batch(x, x_len, F.max_len_wide(x_len))
print("reference decode of %d utterances: %.1f s" % (F.WIDE_B, time.time() - t0))
out = D.pack(ret["predictions"], ret["scores"])
out["enc_sample"] = enc[:, ::7, ::37].numpy()
lab = [sum(1 for e in ret["predictions"][b][0] if int(e)) for b in range(F.WIDE_B)]
print("top-1 labels per utterance: min %d median %d max %d" % (min(lab), sorted(lab)[len(lab) // 2], max(lab)))
np.savez_compressed(os.path.join(HERE, "decode_full_wide.npz"), **out)
print("wrote decode_full_wide.npz", os.path.getsize(os.path.join(HERE, "decode_full_wide.npz")) // 1024, "KiB")
