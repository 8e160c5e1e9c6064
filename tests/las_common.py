from types import SimpleNamespace
import torch

V, H, EMB, C_IN = 41, 32, 12, 64   # output_dim (incl. SOS/EOS ids), rnn_size, embd, encoder-output dim
SOS, EOS, PAD = 39, 40, 41


def opt(attn="mlp"):
    return SimpleNamespace(rnn_size=H, encoder_type="rnn", rnn_type="LSTM", brnn=True, enc_layers=2, dropout=0.0,
                           use_downsampler=False, embd_dim=EMB, num_heads=1, sampling_decoder=False, input_feed=1,
                           dec_layers=2, global_attention=attn, coverage_attn=False, context_gate=None,
                           copy_attn=False)


def inputs():
    g = torch.Generator().manual_seed(99)
    x = torch.randn(23, 1, C_IN, generator=g)
    hyps = [[3, 7, 7, 12], [5], [], [8, 1, 30, 2, 2, 19, 4]]
    return x, hyps


def train_batch():
    """Three utterances of encoder outputs (time-major, zero-padded, lengths sorted as pack_padded_sequence wants them)
    and their targets [SOS] + labels + [EOS], padded with PAD: (src (S,B,C), tgt (L,B,1), lengths (B,) int32)."""
    g = torch.Generator().manual_seed(123)
    lens = [21, 17, 9]
    src = torch.zeros(21, 3, C_IN)
    for b, n in enumerate(lens):
        src[:n, b] = torch.randn(n, C_IN, generator=g)
    labs = [[3, 7, 7, 12, 30], [5, 2], [8, 1, 30, 2]]
    L = max(len(x) for x in labs) + 2
    tgt = torch.full((L, 3, 1), PAD, dtype=torch.long)
    for b, x in enumerate(labs):
        seq = [SOS] + x + [EOS]
        tgt[:len(seq), b, 0] = torch.tensor(seq)
    return src, tgt, torch.tensor(lens, dtype=torch.int32)
