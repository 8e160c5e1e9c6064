"""Generates tests/golden/las_rescore.npz: per-token log-probs of the REFERENCE LAS rescorer
(trainer/model/las.py through decoder/transducer_decoder.py:219-236 `las_rescore`) on seeded
weights, for mlp and general attention.   python tests/golden/make_las_golden.py"""
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
import las_common as LC  # noqa: E402

las, tdec = pika_ref.load_reference("trainer.model.las", "decoder.transducer_decoder")
out = {}
for attn in ("mlp", "general"):
    net = las.Net(LC.opt(attn), LC.C_IN, LC.V, LC.PAD)
    net.load_state_dict(pika_ref.seeded_state_dict(net, 31, scale=0.3))
    net.eval()
    args = SimpleNamespace(las_rescorer=net, las_rescorer_bw=None, bilas_rescorer=None)
    d = tdec.TransducerDecoder(None, 1, 1, args=args)
    x, hyps = LC.inputs()
    for i, h in enumerate(hyps):
        tgt = torch.LongTensor([LC.SOS] + h + [LC.EOS]).unsqueeze(-1).unsqueeze(-1)
        with torch.no_grad():
            out["%s/las/%d" % (attn, i)] = np.array(d.las_rescore(x, tgt))
    print(attn, out["%s/las/0" % attn])
np.savez_compressed(os.path.join(HERE, "las_rescore.npz"), **out)
