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
