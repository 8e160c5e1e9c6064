"""Causal-conv + transformer prediction network
(reference: trainer/model/rnnt_conv_transformer_lm.py:30-87)."""
import torch
import torch.nn as nn

from . import ops
from .modules import TransformerEncoderLayer


class Net(nn.Module):
    def __init__(self, embeddings, output_dim, d_model, num_layers, heads=8, d_ff=2048,
                 dropout=0.1, max_relative_positions=0, max_size=5000):
        super().__init__()
        self.embeddings = embeddings
        self.output_dim = output_dim
        dims = [embeddings.embedding_dim] + [d_model] * (num_layers - 1)
        self.conv = nn.ModuleList([nn.Conv1d(c, d_model, kernel_size=5, padding=4) for c in dims])
        self.transformer = nn.ModuleList(
            [TransformerEncoderLayer(d_model, heads, d_ff, dropout,
                                     max_relative_positions=max_relative_positions)
             for _ in range(num_layers)])
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.linear_out = nn.Linear(d_model, output_dim)
        # strictly-upper-triangular "future" mask, same buffer name/dtype/shape as the reference
        self.register_buffer("mask", torch.triu(torch.ones(1, max_size, max_size, dtype=torch.uint8),
                                                diagonal=1))

    def forward(self, src, softmax=False):
        B, L = src.shape
        pad = src.eq(self.embeddings.padding_idx).unsqueeze(1).expand(B, L, L)
        mask = pad | self.mask[:, :L, :L].bool()  # key j hidden from query i if j>i or src[j] is padding
        # (under the train step's "mixed" arithmetic this network is an island of the two-term mode, backward included:
        # ops.precision_island)
        with ops.precision_island(self.linear_out.weight) as isl:
            out = isl.inp(self.embeddings(src))
            for conv, layer in zip(self.conv, self.transformer):
                out = ops.relu(ops.causal_conv1d(out, conv.weight, conv.bias))
                out = layer(out, mask=mask)
            out = ops.linear(ops.layer_norm(out, self.layer_norm), self.linear_out.weight,
                             self.linear_out.bias)
            out = isl.out(out)
        if softmax:
            out = torch.log_softmax(out, dim=-1)
        return out
