"""Generates tests/golden/las_train.npz: one TRAINING step of the REFERENCE LAS model (trainer/model/las.py
Net.forward :50-90 with the calling convention of trainer/train_las_bmuf_otfaug.py:227-239) and the decoder
cross-entropy of its LASLossCompute (:66-70, 118-123: log_softmax(dec_proj(outputs)) -> NLLLoss(ignore_index = padding,
sum) against target[1:]) on seeded weights: loss, decoder outputs and every parameter gradient.
    python tests/golden/make_las_train_golden.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pika_ref  # noqa: E402
import las_common as LC  # noqa: E402

las = pika_ref.load_reference("trainer.model.las")
out = {}
for attn in ("mlp", "general"):
    net = las.Net(LC.opt(attn), LC.C_IN, LC.V, LC.PAD)
    net.load_state_dict(pika_ref.seeded_state_dict(net, 31, scale=0.3))
    net.train()
    src, tgt, lens = LC.train_batch()
    outputs, _, _, enc_out = net.forward(src, tgt, lens, None, True, True)
    logp = F.log_softmax(net.dec_proj(outputs.view(-1, outputs.size(2))), dim=1)
    loss = F.nll_loss(logp, tgt[1:].contiguous().view(-1), ignore_index=LC.PAD, reduction="sum")
    loss.backward()
    out["%s/loss" % attn] = np.array(loss.item())
    out["%s/outputs" % attn] = outputs.detach().numpy()
    out["%s/enc_out" % attn] = enc_out.detach().numpy()
    for k, p in net.named_parameters():
        out["%s/grad/%s" % (attn, k)] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    print(attn, loss.item(), outputs.shape)
np.savez_compressed(os.path.join(HERE, "las_train.npz"), **out)
