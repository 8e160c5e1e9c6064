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


# The benchmarked LENGTH (BASELINE.json configs[1]: T ~ 1000 input frames -> T' = 240 lattice frames, U = 50 labels) at a
# batch the reference finishes in a minute on CPU: the encoder's attention over 994 / 976 frames, lattice products over
# B * 240 * 51 rows -- the tile shapes `bench.py` runs, B = 4 instead of 32.
LONG = SimpleNamespace(B=4, T_IN=1000, U=50, LENS=[1000, 987, 951, 1000], U_LENS=[50, 47, 50, 41], SEED=SEED + 7)
SHORT = SimpleNamespace(B=B, T_IN=T_IN, U=U, LENS=LENS, U_LENS=U_LENS, SEED=SEED + 1)


def inputs(sc=SHORT):
    g = torch.Generator().manual_seed(sc.SEED)
    x = torch.randn(sc.B, sc.T_IN, D_IN, generator=g)
    lens = torch.tensor(sc.LENS)
    for n in range(sc.B):          # the loader pads with the last frame (otf_utt_loader.py:267-268)
        x[n, sc.LENS[n]:] = x[n, sc.LENS[n] - 1]
    x_len = ((lens - 42) // 4 + ((lens - 42) % 4 != 0).long()).int()      # train_transducer_bmuf_otfaug.py:80-82
    y = torch.randint(1, V, (sc.B, sc.U), generator=g)
    y_len = torch.tensor(sc.U_LENS, dtype=torch.int32)
    for n in range(sc.B):
        y[n, sc.U_LENS[n]:] = V    # padding label
    return x, y, x_len, y_len


def enc_slice(enc):
    return enc[:, ::3, ::17]


def lp_slice(lp):
    return lp[:, ::6, ::4, ::61]


def lp_slice_long(lp):
    return lp[:, ::13, ::7, ::61]
