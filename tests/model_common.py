"""Tiny-model scenario shared by tests/golden/make_model_golden.py (reference side) and
tests/test_model.py (our side): BASELINE.json configs[0] -- 2 transformer layers in the encoder
(6 TDNN layers, nhid 64), V=100, synthetic 80-d fbank spliced to 240."""
from types import SimpleNamespace

import torch

V, H, EMB, NHID, LAYERS, D_IN = 100, 64, 16, 64, 6, 240
B, T_IN, U = 4, 90, 6
SEED = 4242


def make_opt(decoder_type):
    return SimpleNamespace(rnn_size=H, local_rank=0, decoder_type=decoder_type, brnn=False,
                           encoder_type="tdnn", dropout=0.0, enc_layers=2, dec_layers=2,
                           embd_dim=EMB, padding_idx=V)


def build(transducer_mod, encoder_mod, decoder_type):
    """Tiny transducer: the 1024-wide/9-layer encoder hard-coded in transducer.Net is swapped
    for the `encoder_mod.Net(240, 0, 64, tdnn_nhid=64, tdnn_layers=6)` of SURVEY.md 5.6."""
    torch.manual_seed(0)
    opt = make_opt(decoder_type)
    opt_small = SimpleNamespace(**vars(opt))
    opt_small.encoder_type = "rnn"  # build a cheap placeholder encoder, replaced below
    opt_small.enc_layers = 1
    net = transducer_mod.Net(opt_small, D_IN, V)
    net.encoder = encoder_mod.Net(D_IN, 0, H, tdnn_nhid=NHID, tdnn_layers=LAYERS)
    net.pack_seq = False
    for m in net.modules():  # dropout off: parity runs (SURVEY.md 7 "hard parts")
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net


def inputs():
    g = torch.Generator().manual_seed(SEED)
    x = torch.randn(B, T_IN, 80, generator=g) * 4 + 8
    # splice +-1 like loader/otf_utt_loader.py:28-46 (edge frames replicated) -> 240 dims
    left = torch.cat((x[:, :1], x[:, :-1]), 1)
    right = torch.cat((x[:, 1:], x[:, -1:]), 1)
    x = torch.cat((left, x, right), -1)
    x = (x - x.mean(dim=(0, 1))) / 4.0
    y = torch.randint(1, V, (B, U), generator=g)
    y_len = torch.tensor([U, U - 2, 1, U - 1])
    for n in range(B):
        y[n, y_len[n]:] = V  # padding label
    w = torch.randn(B, 1, U + 1, V, generator=g)  # loss weights for the gradient probe
    return x, y, y_len, w
