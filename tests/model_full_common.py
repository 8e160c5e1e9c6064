"""Full-ARCHITECTURE training scenario shared by tests/golden/make_model_full_golden.py (reference side) and
tests/test_model_full.py: the model of BASELINE.json configs[1] (TDNN-Transformer encoder 1024 wide / 9 layers,
conv-transformer prediction net, H = 1024, V = 5000) in TRAIN mode (BatchNorm on batch statistics, dropout as the
identity) on B = 3 utterances of 3.9-4.2 s, T' = 95 lattice frames, U = 12.

Weights: the tests' seeded generator at ~1/sqrt(fan-in) (tests/decode_full_common.py explains why not 0.1); fc2 is
sharpened so that the lattice posteriors are not uniform and the RNN-T cost depends on the encoder output."""
from types import SimpleNamespace

import torch

V, H, EMB, D_IN = 5000, 1024, 100, 240
B, T_IN, U = 3, 420, 12
LENS = [420, 406, 388]
U_LENS = [12, 9, 11]
SEED, SCALE = 929, 0.02


def opt(decoder_type="transformer"):
    """decoder_type "rnn": the 2-layer LSTM prediction network every shipped recipe trains (egs/train_transducer_bmuf_otfaug.sh:32,
    trainer/model/transducer.py:55-61)."""
    return SimpleNamespace(rnn_size=H, local_rank=0, decoder_type=decoder_type, brnn=False, encoder_type="tdnn",
                           dropout=0.0, enc_layers=4, dec_layers=2, embd_dim=EMB, padding_idx=V)


def build(transducer_mod, seeded_state_dict, decoder_type="transformer"):
    torch.manual_seed(0)
    net = transducer_mod.Net(opt(decoder_type), D_IN, V)
    net.load_state_dict(seeded_state_dict(net, SEED, scale=SCALE))
    with torch.no_grad():
        net.fc2.weight *= 25.0
        net.fc2.bias[0] += 2.0
    for m in net.modules():     # the encoder's transformer layers hard-code dropout 0.2 (rnnt_tdnn_transformer.py:62-65)
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net.train()


def inputs():
    g = torch.Generator().manual_seed(SEED + 1)
    x = torch.randn(B, T_IN, D_IN, generator=g)
    lens = torch.tensor(LENS)
    for n in range(B):          # the loader pads with the last frame (otf_utt_loader.py:267-268)
        x[n, LENS[n]:] = x[n, LENS[n] - 1]
    x_len = ((lens - 42) // 4 + ((lens - 42) % 4 != 0).long()).int()      # train_transducer_bmuf_otfaug.py:80-82
    y = torch.randint(1, V, (B, U), generator=g)
    y_len = torch.tensor(U_LENS, dtype=torch.int32)
    for n in range(B):
        y[n, U_LENS[n]:] = V    # padding label
    return x, y, x_len, y_len


def enc_slice(enc):
    return enc[:, ::3, ::17]


def lp_slice(lp):
    return lp[:, ::6, ::4, ::61]
