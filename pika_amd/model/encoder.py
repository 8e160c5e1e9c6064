"""TDNN-Transformer encoder (reference: trainer/model/rnnt_tdnn_transformer.py:27-89)."""
import torch
import torch.nn as nn

from . import ops
from .modules import TransformerEncoderLayer


class Net(nn.Module):
    """fc_in+ReLU+BN, `tdnn_layers` time-delay layers (3 taps; dilation 1 for the first three,
    3 afterwards; the last one also strides by 4) each followed by ReLU+BN, a transformer layer
    after every third TDNN layer (heads 16/16/8, d_ff = 4*nhid, dropout 0.2, no mask, no
    positional encoding), then BN + fc_out."""

    HEADS = (16, 16, 8)

    def __init__(self, input_dim, input_ctx, output_dim, tdnn_nhid, tdnn_layers, bn_dim=0):
        super().__init__()
        assert tdnn_layers > 4
        self.input_dim, self.output_dim, self.tdnn_nhid = input_dim, output_dim, tdnn_nhid
        self.filter_size = 3
        self.fc_in = nn.Linear(input_dim, tdnn_nhid)
        self.bn_in = nn.BatchNorm1d(tdnn_nhid)

        def td(dil, stride=1):  # parameters kept in the reference's Conv2d container/shape
            return nn.Conv2d(1, tdnn_nhid, kernel_size=(self.filter_size, tdnn_nhid),
                             stride=(stride, 1), dilation=(dil, 1))
        layers = [td(1) for _ in range(3)] + [td(3) for _ in range(tdnn_layers - 4)] + [td(3, 4)]
        self.hidden_conv = nn.ModuleList(layers)
        self.hidden_bn = nn.ModuleList([nn.BatchNorm1d(tdnn_nhid) for _ in range(tdnn_layers)])
        self.transformer = nn.ModuleList(
            [TransformerEncoderLayer(tdnn_nhid, h, tdnn_nhid * 4, 0.2, max_relative_positions=0)
             for h in self.HEADS])
        self.bn_final = nn.BatchNorm1d(tdnn_nhid)
        self.fc_out = nn.Linear(tdnn_nhid, output_dim)

    def forward(self, x, frame_offset=0, valid_frames=None):
        """valid_frames (training on a HIP device only; pika_amd/train_graph.py): a (1,) int32 DEVICE tensor V <= T -- the
        time axis of x is padded beyond its V frames of data up to a fixed shape.  Frames beyond V then take no part in
        anything the reference computes on the unpadded batch: BatchNorm statistics / row counts / backward run over the
        data rows of every layer (V shrinks with the time-delay layers: pika_bn_valid_t), padding frames are masked as
        attention keys, and everything else is row-local or a valid convolution (data rows never read padding rows); the
        outputs of the first (V - 39) // 4 frames and every parameter gradient equal those of the unpadded batch."""
        B = x.size(0)
        C = self.tdnn_nhid
        valid = ops.valid_rows if valid_frames is not None else (lambda *a: _NoValid())
        sub, div = 0, 1         # data frames at the current layer = (V - sub) // div
        fuse = ops.fused_relu_bn_ok(x, self.bn_in)   # ReLU fwd in the GEMM epilogue, bwd in the BN backward
        h = ops.linear(x, self.fc_in.weight, self.fc_in.bias, relu=2 if fuse else 1)
        n_layers = len(self.hidden_conv)
        # a BatchNorm output that only feeds the next time-delay GEMM is produced in bf16 (bf16 mode)
        feeds_tdnn = fuse and ops.tdnn_bn_ok(h, self.hidden_conv[0].weight, self.hidden_bn[0])
        with valid(valid_frames, h.shape[1], sub, div):
            h = ops.batch_norm(h.reshape(-1, C), self.bn_in, relu_input=fuse, mfma_only=feeds_tdnn).view(B, -1, C)
        for i, (conv, bn) in enumerate(zip(self.hidden_conv, self.hidden_bn)):
            to_transformer = (i + 1) % 3 == 0
            span, stride = conv.dilation[0] * (self.filter_size - 1), conv.stride[0]
            t_out = (h.shape[1] - span - 1) // stride + 1
            # data frames after a valid convolution: (v - span - 1) // stride + 1 = (v - span - 1 + stride) // stride
            assert div == 1, "only the last time-delay layer strides"
            sub, div = sub + span + 1 - stride, stride
            if ops.tdnn_bn_ok(h, conv.weight, bn):
                nxt_ok = (not to_transformer and i + 1 < n_layers)
                with valid(valid_frames, t_out, sub, div):
                    h = ops.tdnn_bn(h, conv, bn, mfma_only=nxt_ok)
            else:
                fuse = ops.fused_relu_bn_ok(h, bn)
                h = ops.tdnn(h.float() if h.dtype != torch.float32 else h, conv.weight, conv.bias,
                             conv.dilation[0], conv.stride[0], relu=2 if fuse else 1)
                with valid(valid_frames, t_out, sub, div):
                    h = ops.batch_norm(h.reshape(-1, C), bn, relu_input=fuse).view(B, -1, C)
            if to_transformer:
                mask = None
                if valid_frames is not None:        # padding frames are no keys (the reference attends over the data)
                    v = torch.div(valid_frames.to(torch.long) - sub, div, rounding_mode="floor")
                    T_l = h.shape[1]
                    mask = (torch.arange(T_l, device=h.device) >= v).view(1, 1, T_l).expand(B, T_l, T_l)
                h = self.transformer[i // 3](h, mask=mask)
        with valid(valid_frames, h.shape[1], sub, div):
            h = ops.batch_norm(h.reshape(-1, C), self.bn_final)
        h = ops.linear(h, self.fc_out.weight, self.fc_out.bias).view(B, -1, self.output_dim)
        return h[:, frame_offset:, :]


class _NoValid(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False
