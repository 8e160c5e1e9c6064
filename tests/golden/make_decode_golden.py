"""Generates tests/golden/decode_tiny.npz by running the REFERENCE decoder
(decoder/transducer_decoder.py + decoder/beam_transducer.py, imported from /root/reference under
the shims of oracle/pika_ref.py) on the tiny model of tests/model_common.py, CPU fp32.
    python tests/golden/make_decode_golden.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pika_ref  # noqa: E402
import model_common as C  # noqa: E402
import decode_common as D  # noqa: E402

transducer, encoder, tdec, beam_mod = pika_ref.load_reference(
    "trainer.model.transducer", "trainer.model.rnnt_tdnn_transformer",
    "decoder.transducer_decoder", "decoder.beam_transducer")

out = {}
for dec in ("transformer", "rnn"):
    net = C.build(transducer, encoder, dec)
    net.load_state_dict(pika_ref.seeded_state_dict(net, C.SEED))
    D.tweak(net)
    net.eval()
    x, x_len = D.inputs()
    for name, cfg in D.SCENARIOS.items():
        args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None,
                               nonblk_reward=0.0)
        d = tdec.TransducerDecoder(net, batch_size=x.shape[0], beam_size=cfg["beam"],
                                   n_best=cfg["n_best"], blk=0,
                                   global_scorer=beam_mod.GlobalScorer(), sm_scale=cfg["sm_scale"],
                                   cuda=False, beam_prune=True, args=args)
        with torch.no_grad():
            ret, enc = d.decode_batch(x, x_len, D.max_len(cfg, x_len))
        preds, scores = ret["predictions"], ret["scores"]
        flat = D.pack(preds, scores)
        for k, v in flat.items():
            out["%s/%s/%s" % (dec, name, k)] = v
        print(dec, name, "first hyp:", [int(e) for e in preds[0][0]][:24], float(scores[0][0]))
np.savez_compressed(os.path.join(HERE, "decode_tiny.npz"), **out)
print("wrote decode_tiny.npz")
