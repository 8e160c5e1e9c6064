"""Generates tests/golden/las_train.npz: one TRAINING step of the REFERENCE LAS model (trainer/model/las.py
Net.forward :50-90 with the calling convention of trainer/train_las_bmuf_otfaug.py:227-239) and the decoder
cross-entropy of its LASLossCompute (:66-70, 118-123: log_softmax(dec_proj(outputs)) -> NLLLoss(ignore_index = padding,
sum) against target[1:]) on seeded weights: loss, decoder outputs and every parameter gradient -- for the default call
(encoder + decoder), for decoder pre-training (enable_enc False, :92-116), for the encoder-only call with the script's CTC
branch (enable_dec False, :98-131 of the script) and for a call that continues from a returned decoder state.
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


def record(tag, net, loss, extra):
    loss.backward()
    out["%s/loss" % tag] = np.array(loss.item())
    for k, v in extra.items():
        out["%s/%s" % (tag, k)] = v.detach().numpy()
    for k, p in net.named_parameters():
        out["%s/grad/%s" % (tag, k)] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    print(tag, loss.item())


# The other modes the training script drives (trainer/train_las_bmuf_otfaug.py:193-239, 98-131):
def fresh():
    net = las.Net(LC.opt("mlp"), LC.C_IN, LC.V, LC.PAD)
    net.load_state_dict(pika_ref.seeded_state_dict(net, 31, scale=0.3))
    return net.train()


src, tgt, lens = LC.train_batch()
# (1) --pretrain_decoder: enable_enc False, the decoder as a language model; cross-entropy as above
net = fresh()
outputs, a, b, c = net.forward(src, tgt, lens, None, True, False)
assert a is None and b is None and c is None
logp = F.log_softmax(net.dec_proj(outputs.view(-1, outputs.size(2))), dim=1)
record("pretrain_dec", net, F.nll_loss(logp, tgt[1:].contiguous().view(-1), ignore_index=LC.PAD, reduction="sum"),
       {"outputs": outputs})
# (2) dec_loss_scale 0: enable_dec False, only the encoder runs; the script's CTC branch on enc_proj(enc_out) (:66-83)
net = fresh()
o, a, b, enc_out = net.forward(src, tgt, lens, None, False, True)
assert o is None and a is None and b is None
L, B_, _ = enc_out.shape
pout = net.enc_proj(enc_out.view(-1, enc_out.size(2))).view(L, B_, -1)
t2 = tgt.view(tgt.size(0), -1).transpose(0, 1)
mask = torch.lt(t2, LC.PAD) & torch.gt(t2, 1)
label = t2[mask].int()
label_size = mask.int().sum(1)
ctc = torch.nn.CTCLoss()(pout, label, torch.as_tensor(lens).int(), label_size)
record("enc_only_ctc", net, ctc, {"enc_out": enc_out})
# (3) a carried decoder state: the second half of the targets continued from the state the first half returned
net = fresh()
half = (tgt.size(0) - 1) // 2
tgt_a, tgt_b = tgt[:half + 1], tgt[half:]
out_a, _, st, _ = net.forward(src, tgt_a, lens, None, True, True)
out_b, _, st2, _ = net.forward(src, tgt_b, lens, st, True, True)
outputs = torch.cat([out_a, out_b], 0)
logp = F.log_softmax(net.dec_proj(outputs.view(-1, outputs.size(2))), dim=1)
tg = torch.cat([tgt_a[1:], tgt_b[1:]], 0)
record("carried_state", net, F.nll_loss(logp, tg.contiguous().view(-1), ignore_index=LC.PAD, reduction="sum"),
       {"outputs": outputs})
np.savez_compressed(os.path.join(HERE, "las_train.npz"), **out)
