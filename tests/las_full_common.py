"""Full-WIDTH LAS rescoring scenario shared by tests/golden/make_las_full_golden.py (reference side) and
tests/test_las_full.py: the rescorer of BASELINE.json configs[4] as bench.py builds it (2-layer BLSTM 1024 = 512 per
direction, 2-layer input-feed LSTM decoder 1024, mlp attention, embd 100, output_dim = V + 2 with SOS = V, EOS = V + 1,
PAD = V + 2) on 4 utterances of ragged length x 8 n-best entries that share prefixes the way real n-best lists do
(substitutions, deletions, insertions, an entry that is a prefix of another, a duplicate, an empty one).

Weights: the tests' seeded generator at ~1/sqrt(fan-in); dec_proj sharpened so that token log-probs spread over
several nats (a near-uniform posterior would hide errors of the decoder state)."""
from types import SimpleNamespace

import numpy as np
import torch

V, H, EMB, C_IN = 5000, 1024, 100, 1024
SOS, EOS, PAD = V, V + 1, V + 2
OUT = V + 2
LENS = [61, 57, 48, 52]
SEED_FW, SEED_BW, SCALE = 4101, 4102, 0.03
NBEST = 8


def opt():
    return SimpleNamespace(rnn_size=H, encoder_type="rnn", rnn_type="LSTM", brnn=True, enc_layers=2, dropout=0.0,
                           use_downsampler=False, embd_dim=EMB, num_heads=1, sampling_decoder=False, input_feed=1,
                           dec_layers=2, global_attention="mlp", coverage_attn=False, context_gate=None,
                           copy_attn=False)


def build(las_mod, seeded_state_dict, seed):
    torch.manual_seed(0)
    net = las_mod.Net(opt(), C_IN, OUT, PAD)
    net.load_state_dict(seeded_state_dict(net, seed, scale=SCALE))
    with torch.no_grad():
        net.dec_proj.weight *= 30.0
    return net.eval()


def inputs():
    """(src (S,B,C) zero-padded encoder outputs, lengths, hyps[b] = list of NBEST label lists)."""
    g = torch.Generator().manual_seed(4100)
    src = torch.zeros(max(LENS), len(LENS), C_IN)
    for b, n in enumerate(LENS):
        src[:n, b] = torch.randn(n, C_IN, generator=g)
    rng = np.random.default_rng(4100)
    hyps = []
    for b in range(len(LENS)):
        base = [int(v) for v in rng.integers(1, V, size=int(rng.integers(14, 23)))]
        row = [list(base)]
        s = list(base); s[len(s) // 2] = int(rng.integers(1, V)); row.append(s)              # substitution
        d = list(base); del d[len(d) // 3]; row.append(d)                                     # deletion
        i = list(base); i.insert(2 * len(i) // 3, int(rng.integers(1, V))); row.append(i)     # insertion
        row.append(list(base[:len(base) // 2]))                                               # a prefix of entry 0
        row.append(list(base))                                                                # a duplicate
        t = list(base); t[-1] = int(rng.integers(1, V)); t.append(int(rng.integers(1, V))); row.append(t)
        row.append([] if b % 2 == 0 else [int(rng.integers(1, V))])                           # empty / one label
        assert len(row) == NBEST
        hyps.append(row)
    return src, LENS, hyps
