"""Functional ops used by the model forward passes.

Dispatch rule: GPU tensors -> HIP kernels of libpika_amd.so where one exists (listed in
DESIGN.md "kernel inventory"); plain library GEMMs go to hipBLASLt through torch.matmul.
CPU tensors (module-structure tests, gloo plumbing) use the equivalent torch ops.
"""
import math

import os

import torch
import torch.nn.functional as F


def _hip(x):
    return x.is_cuda


def _bf16_mode():
    """Modes whose activations between MFMA products are bf16: "bf16" (one plane) and "mixed" (two planes, hipops.Pair)."""
    from .. import gemm as G
    return G.bf16_backward()


def _mixed():
    from .hipops import _mixed as m
    return m()


def _pair(ret):
    """(hi, lo) returned by an autograd Function with a two-term output -> hipops.Pair; tensors pass through."""
    if isinstance(ret, tuple):
        from .hipops import Pair
        return Pair(ret[0], ret[1])
    return ret


def _hi_lo(x):
    """(tensor, lo plane or None) of an activation that may be a hipops.Pair."""
    from .hipops import Pair
    return (x.hi, x.lo) if isinstance(x, Pair) else (x, None)


def _bf16_fast():
    """bf16 arithmetic mode with the fused / bf16-activation paths enabled (PIKA_NO_FUSED unset)."""
    from .hipops import _fused
    return _bf16_mode() and _fused()


# ---- a sub-network in another arithmetic than the rest of the step ---------------------------------------------------
# "mixed" runs every backward product on one bf16 term.  For the encoder and the joint that is the benchmarked trade (their
# gradients stay within 2-4e-2 of the reference's); the conv-transformer prediction network is 1632 rows -- a few per cent of
# a step -- and its query / key gradients are the small remainder of a nearly shift-invariant softmax: on one bf16 term they
# were 0.11 off.  It therefore runs as an island of the two-term mode ("bf16x3": fp32 tensors between products, two-term
# products forward AND backward).  The mode is a process global that kernels are picked by at call time, so the island is
# entered / left by two identity autograd nodes: autograd runs nodes in reverse order of creation, every node of the
# prediction network was created after every node of the encoder, so between `island_enter`'s backward (the first node of
# the island to run) and `island_leave`'s (the last) no node from outside the island runs.
_ISLAND_STACK = []


class _IslandEnter(torch.autograd.Function):
    """Identity on the island's OUTPUT: its backward runs first and switches the arithmetic."""

    @staticmethod
    def forward(ctx, x, mode):
        ctx.mode = mode
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        from .. import gemm as G
        _ISLAND_STACK.append(G.PRECISION)
        G.PRECISION = ctx.mode
        return g, None


class _IslandLeave(torch.autograd.Function):
    """Identity on the island's INPUT: its backward runs last and restores the arithmetic."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        from .. import gemm as G
        if _ISLAND_STACK:
            G.PRECISION = _ISLAND_STACK.pop()
        return g


class precision_island(object):
    """with precision_island(x) as isl:  h = isl.inp(first differentiable tensor);  ...;  y = isl.out(result)
    Active (mode "bf16x3") only for a training call on a HIP device under the "mixed" arithmetic; otherwise a no-op."""

    def __init__(self, probe):
        from .. import gemm as G
        precision_island.recover()
        self.left = False
        self.mode = "bf16x3" if (G.PRECISION == "mixed" and probe.is_cuda and torch.is_grad_enabled()) else None
        self.old = None

    def __enter__(self):
        if self.mode is not None:
            from .. import gemm as G
            self.old, G.PRECISION = G.PRECISION, self.mode
        return self

    @staticmethod
    def recover():
        """A backward pass that entered an island and never left it (an exception inside it, a pass cut short by
        autograd.grad(inputs=...)) leaves the process in the island's arithmetic: the next forward puts the outermost
        saved mode back.  Called at the start of every island forward and of every graphed / eager training forward."""
        if _ISLAND_STACK:
            from .. import gemm as G
            G.PRECISION = _ISLAND_STACK[0]
            del _ISLAND_STACK[:]

    def __exit__(self, *exc):
        if self.mode is not None:
            from .. import gemm as G
            G.PRECISION = self.old
        return False

    def inp(self, x):
        if self.mode is None:
            return x
        if not x.requires_grad:
            # frozen embeddings: the Leave node must still exist (and run last), or Enter's switch is never undone --
            # a throw-away leaf gives it a gradient edge
            x = x.detach().requires_grad_(True)
        self.left = True
        return _IslandLeave.apply(x)

    def out(self, y):
        # Enter only ever together with Leave
        return _IslandEnter.apply(y, self.mode) if (self.mode is not None and self.left and y.requires_grad) else y


def tdnn_bn_ok(x, conv_weight, bn):
    """bn(relu(tdnn(x))) can run as ONE node (hipops.TdnnBnFn): bf16 fast mode, training statistics, widths
    the direct-to-LDS kernels take, enough rows to be worth a bf16 copy."""
    N, _, taps, C = conv_weight.shape
    return (_hip(x) and _bf16_fast() and fused_relu_bn_ok(x, bn) and C % 64 == 0 and N % 64 == 0
            and x.dtype in (torch.float32, torch.bfloat16) and x.numel() >= (1 << 21))


def tdnn_bn(x, conv, bn, mfma_only=False):
    from .hipops import TdnnBnFn
    N, _, taps, C = conv.weight.shape
    mom = _bn_momentum(bn)
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    x, x_lo = _hi_lo(x)
    return _pair(TdnnBnFn.apply(x, conv.weight.reshape(N, taps * C), conv.bias, taps, conv.dilation[0], conv.stride[0], 0,
                                bn.weight, bn.bias, rm, rv, bn.eps, mom, bool(mfma_only), x_lo,
                                bool(mfma_only) and _mixed()))


def _gemm_ok(*dims):
    return all(d % 4 == 0 for d in dims)


def linear(x, weight, bias=None, relu=False, out_bf16=False):
    """nn.Linear semantics (+ optional fused ReLU epilogue); MFMA GEMM kernel on the GPU.
    out_bf16 (bf16 mode only): the result only feeds MFMA products."""
    if _hip(x) and _gemm_ok(weight.shape[1]):
        from .hipops import LinearFn
        x, x_lo = _hi_lo(x)
        return _pair(LinearFn.apply(x, weight, bias, relu, bool(out_bf16 and _bf16_mode()), x_lo,
                                    bool(out_bf16) and _mixed() and weight.shape[0] % 64 == 0))
    y = F.linear(x.to(weight.dtype), weight, bias)
    return F.relu(y) if relu else y


def relu(x):
    return F.relu(x)


def fused_relu_bn_ok(x, bn):
    """True when `bn(relu(gemm(x)))` can run as GEMM(+ReLU epilogue) -> BatchNorm kernels whose
    backward also applies the ReLU mask (no separate ReLU passes)."""
    return _hip(x) and (bn.training or not bn.track_running_stats) and bn.affine and \
        torch.is_grad_enabled()


def valid_rows(t_valid, rows_per_batch, sub, div):
    """Context for the BatchNorm launches of a layer whose time axis is padded (hipops.valid_rows)."""
    from .hipops import valid_rows as ctx
    return ctx(t_valid, rows_per_batch, sub, div)


def batch_norm(x2d, bn, relu_input=False, mfma_only=False):
    """BatchNorm1d over rows of a (M,C) matrix with the module's buffers (train: batch stats).
    relu_input: x2d is a ReLU output whose backward mask this op's backward must apply.
    mfma_only: the result only feeds an MFMA product (bf16 mode: produced in bf16)."""
    train_stats = bn.training or not bn.track_running_stats
    if _hip(x2d) and train_stats and bn.affine and x2d.shape[1] % 4 == 0 and x2d.dtype == torch.float32:
        from .hipops import BatchNormFn
        mom = _bn_momentum(bn)
        rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
        narrow = bool(mfma_only and _bf16_fast() and x2d.shape[1] % 64 == 0)
        return _pair(BatchNormFn.apply(x2d, bn.weight, bn.bias, rm, rv, bn.eps, mom, relu_input, narrow,
                                       narrow and _mixed()))
    return F.batch_norm(x2d, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                        bn.training or not bn.track_running_stats,
                        _bn_momentum(bn), bn.eps)


def _bn_momentum(bn):
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if bn.momentum is None:
            return 1.0 / float(bn.num_batches_tracked)
    return 0.0 if bn.momentum is None else bn.momentum


def layer_norm(x, ln, mfma_only=False, with_skip=False):
    """nn.LayerNorm over the last dim.  mfma_only: every consumer of the result is an MFMA product
    (projection / feed-forward GEMMs), so in the bf16 arithmetic mode it is produced in bf16.
    with_skip: returns (LN(x), x') where x' is x for the caller's skip connection around the LN -- on the HIP path an alias
    whose gradient is added to the LN's input gradient inside the backward kernel (hipops.LayerNormFn)."""
    C = x.shape[-1]
    if (_hip(x) and x.dtype == torch.float32 and len(ln.normalized_shape) == 1 and ln.weight is not None
            and ln.bias is not None and C % 4 == 0 and C <= 2048):
        from .hipops import LayerNormFn
        narrow = bool(mfma_only and _bf16_mode() and C % 64 == 0)
        if with_skip and x.requires_grad and torch.is_grad_enabled():
            ret = LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps, narrow, narrow and _mixed(), True)
            return _pair(ret[:-1] if len(ret) > 2 else ret[0]), ret[-1]
        y = _pair(LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps, narrow, narrow and _mixed()))
        return (y, x) if with_skip else y
    y = F.layer_norm(x, ln.normalized_shape, ln.weight, ln.bias, ln.eps)
    return (y, x) if with_skip else y


def dropout(x, p, training):
    return F.dropout(x, p, training) if (training and p > 0.0) else x


def tdnn(x, weight, bias, dilation, stride, relu=0):
    """Time-delay layer: y[b,t,n] = sum_j sum_c W[n,0,j,c] * x[b, t*stride + j*dilation, c] + bias[n].

    x (B,T,C); weight is the reference's Conv2d weight (N,1,taps,C)
    (rnnt_tdnn_transformer.py:44-57 builds it as a (taps x C) 2-d convolution).  Computed as ONE
    GEMM with K = taps*C over time-shifted views -- never as a convolution."""
    N, _, taps, C = weight.shape
    B, T, _ = x.shape
    if _hip(x) and _gemm_ok(C):
        from .hipops import TimeDelayFn
        return TimeDelayFn.apply(x, weight.reshape(N, taps * C), bias, taps, dilation, stride, 0, relu)
    span = T - dilation * (taps - 1)
    t_out = (span - 1) // stride + 1
    cols = [x[:, j * dilation: j * dilation + (t_out - 1) * stride + 1: stride, :] for j in range(taps)]
    a = torch.cat(cols, dim=-1)  # (B, t_out, taps*C)
    y = F.linear(a, weight.reshape(N, taps * C), bias)
    return F.relu(y) if relu else y


def causal_conv1d(x, weight, bias):
    """y[b,t,n] = sum_j sum_c W[n,c,j] * x[b, t - (k-1) + j, c] + bias (zeros left of t=0).

    x (B,T,C), weight (N,C,k): the reference pads k-1 both sides and trims the right
    (rnnt_conv_transformer_lm.py:36-45,73)."""
    N, C, k = weight.shape
    B, T, _ = x.shape
    if _hip(x) and _gemm_ok(C):
        from .hipops import TimeDelayFn
        w2 = weight.permute(0, 2, 1).reshape(N, k * C)  # tap-major columns
        return TimeDelayFn.apply(x, w2, bias, k, 1, 1, k - 1, False)
    xp = F.pad(x, (0, 0, k - 1, 0))
    a = torch.cat([xp[:, j: j + T, :] for j in range(k)], dim=-1)  # (B,T,k*C), tap-major
    w = weight.permute(0, 2, 1).reshape(N, k * C)
    return F.linear(a, w, bias)


def attention(q, k, v, heads, mask, p_drop, training):
    """Multi-head scaled dot-product attention on (B,T,H*D) projections; returns (B,Tq,H*D).

    Matches multi_headed_attn.py:199-231: q scaled by 1/sqrt(D) BEFORE q.k^T, scores in fp32,
    masked_fill(mask, -1e18), softmax, dropout on the probabilities."""
    B, Tq, HD = q.shape
    Tk = k.shape[1]
    D = HD // heads
    if _hip(q):
        from .hipops import AttentionFn, attention_ok, attention_infer_ok, attention_infer_two_term
        if attention_infer_ok(q, k, v, heads, mask):
            return attention_infer_two_term(q, k, v, heads, mask)
        if attention_ok(q, k, v, heads, mask) and not _mixed():
            drop = p_drop if training else 0.0
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if drop > 0 else 0  # CPU generator
            return AttentionFn.apply(q, k, v, heads, drop, seed, mask)
    qh = (q / math.sqrt(D)).view(B, Tq, heads, D).transpose(1, 2)
    kh = k.view(B, Tk, heads, D).transpose(1, 2)
    vh = v.view(B, Tk, heads, D).transpose(1, 2)
    scores = torch.matmul(qh, kh.transpose(2, 3)).float()
    if mask is not None:
        scores = scores.masked_fill(mask.unsqueeze(1), -1e18)
    attn = torch.softmax(scores, dim=-1).to(q.dtype)
    ctx = torch.matmul(dropout(attn, p_drop, training), vh)
    return ctx.transpose(1, 2).contiguous().view(B, Tq, HD)


def self_attention_infer_ok(x, heads, mask):
    """Inference self-attention of a two-term decode mode on ONE packed projection (hipops.attention_infer_packed)."""
    if not _hip(x) or x.shape[-1] % 8:
        return False
    from .hipops import attention_infer_ok
    return attention_infer_ok(x, x, x, heads, mask)


def self_attention_infer(x, w_qkv, b_qkv, heads, mask=None):
    from .hipops import attention_infer_packed
    return attention_infer_packed(linear(x, w_qkv, b_qkv), heads, mask)


def feed_forward_applies(x, w_1, w_2):
    if not _hip(x):
        return False
    from .hipops import feed_forward_ok
    return feed_forward_ok(x, w_1.weight, w_2.weight)


def _seed(p):
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0 else 0   # CPU generator: no device sync


def feed_forward(xn, w_1, w_2, p_drop, residual=None, p_residual=0.0):
    """w_2(dropout(relu(w_1(xn)))) through the fused bf16-hidden path, or None when it does not apply
    (CPU, parity mode, odd widths): the caller then runs the plain chain.  With `residual` the result is
    dropout_{p_residual}(that) + residual, fused into the second GEMM."""
    if not _hip(xn):
        return None
    from .hipops import FeedForwardFn, feed_forward_ok
    if not feed_forward_ok(xn, w_1.weight, w_2.weight):
        return None
    xn, x_lo = _hi_lo(xn)
    if residual is not None and residual.dtype == torch.float32:
        return FeedForwardFn.apply(xn, w_1.weight, w_1.bias, w_2.weight, w_2.bias, p_drop, _seed(p_drop),
                                   residual, p_residual, _seed(p_residual), x_lo)
    y = FeedForwardFn.apply(xn, w_1.weight, w_1.bias, w_2.weight, w_2.bias, p_drop, _seed(p_drop), None, 0.0, 0, x_lo)
    return y if residual is None else dropout(y, p_residual, p_residual > 0) + residual


def linear_dropout_residual(x, lin, residual, p_drop):
    """dropout_p(lin(x)) + residual; one GEMM when x is a bf16 activation (see LinearDropoutResidualFn)."""
    if _hip(x):
        from .hipops import LinearDropoutResidualFn, linear_dropout_residual_ok
        if linear_dropout_residual_ok(x, lin.weight, residual):
            x, x_lo = _hi_lo(x)
            return LinearDropoutResidualFn.apply(x, lin.weight, lin.bias, residual, p_drop, _seed(p_drop), x_lo)
    return dropout(linear(x, lin.weight, lin.bias), p_drop, p_drop > 0) + residual


def self_attention_packed_ok(x, heads, mask):
    if not _hip(x):
        return False
    from .hipops import attention_ok
    return attention_ok(x, x, x, heads, mask)


def self_attention_packed(x, wq, bq, wk, bk, wv, bv, heads, p_drop, training, mask=None):
    """Self-attention context from the layer input: q, k, v projections as ONE GEMM (weights concatenated
    per call; autograd splits the weight gradient back), then the fused attention core on the packed
    result.  Same arithmetic as three separate nn.Linear calls.  "mixed" mode: the projection is written as two bf16
    planes and the attention forward runs in two-term arithmetic (hipops.PackedAttentionFn)."""
    from .hipops import PackedAttentionFn
    w = torch.cat([wq, wk, wv], 0)
    b = torch.cat([bq, bk, bv], 0)
    qkv, qkv_lo = _hi_lo(linear(x, w, b, out_bf16=True))
    drop = p_drop if training else 0.0
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if drop > 0 else 0
    return _pair(PackedAttentionFn.apply(qkv, heads, drop, seed, qkv_lo, mask))


def joint(enc, pred, fc1, fc_gate, fc2, log_softmax=True, scale=1.0, labels=None):
    """Gated joint network over the full (T,U) lattice.

    enc (B,T,H), pred (B,U,H) -> (B,T,U,V).  Reference (transducer.py:98-111) concatenates the
    expanded tensors into (B,T,U,2H) and runs fc1/fc_gate on it (103 GFLOP/utt at config 2);
    here the 2H-wide weights are split into their encoder and prediction halves, applied to
    (B,T,H) and (B,U,H) separately and broadcast-added (1.2 GFLOP/utt), which is the same
    affine map."""
    H = enc.shape[-1]
    w1e, w1p = fc1.weight[:, :H], fc1.weight[:, H:]
    wge, wgp = fc_gate.weight[:, :H], fc_gate.weight[:, H:]
    if _hip(enc) and _gemm_ok(H, fc2.weight.shape[1]):
        from .hipops import GateFn, JointOutFn, LogSoftmaxFn, joint_out_ok
        e1 = linear(enc, w1e.contiguous(), fc1.bias)
        p1 = linear(pred, w1p.contiguous())
        eg = linear(enc, wge.contiguous(), fc_gate.bias)
        pg = linear(pred, wgp.contiguous())
        h = GateFn.apply(e1, p1, eg, pg)
        lazy_on = None
        if log_softmax and h.dim() == 4 and fc2.bias is not None and fc2.weight.shape[0] % 4 and \
                joint_out_ok(h, fc2.weight, units=(fc2.weight.shape[0] + 3) & ~3):
            # An output layer whose size is not a multiple of four: padded to the next one HERE, with a bias of -6e4 on the
            # padding units (probability zero: log-sum-exp, costs and gradients are those of the layer itself), so that it
            # takes the same kernels as any other; the lazy output stands for the layer's own columns.  (Without the lazy
            # output -- PIKA_LAZY_LOGPROBS=0 -- such a layer runs the plain chain below.)
            from ..rnnt import _lazy_enabled
            lazy_on = _lazy_enabled() and os.environ.get("PIKA_LAZY_LOGPROBS", "1") != "0"
            N = fc2.weight.shape[0]
            if lazy_on:
                w4 = F.pad(fc2.weight, (0, 0, 0, (-N) % 4))
                b4 = F.pad(fc2.bias, (0, (-N) % 4), value=-6.0e4)
                lp = JointOutFn.apply(h, w4, b4, scale, True, labels, N)
                lp._pika_lazy_grad_ok = True
                return lp
        if log_softmax and joint_out_ok(h, fc2.weight):
            from ..rnnt import _lazy_enabled
            # the log-softmax pass runs only if something other than this package's RNN-T loss needs the values
            # (pika_amd.rnnt.LazyLogProbs), and the loss may hand back its gradient as a tensor that is only written
            # if something other than JointOutFn looks at it (pika_amd.rnnt.LazyDenseGrad)
            # labels (B,U), optional: the label of every lattice column, which lets the product keep the lattice in 16 bits
            # (the two logits per row the RNN-T loss reads leave its epilogue in fp32; JointOutFn.forward)
            lp = JointOutFn.apply(h, fc2.weight, fc2.bias, scale,
                                  _lazy_enabled() and os.environ.get("PIKA_LAZY_LOGPROBS", "1") != "0", labels)
            lp._pika_lazy_grad_ok = True
            return lp
        out = linear(h, fc2.weight, fc2.bias)
        if log_softmax:
            out = LogSoftmaxFn.apply(out, scale)
        return out
    e1 = F.linear(enc, w1e, fc1.bias)
    p1 = F.linear(pred, w1p)
    eg = F.linear(enc, wge, fc_gate.bias)
    pg = F.linear(pred, wgp)
    h = torch.tanh(e1.unsqueeze(2) + p1.unsqueeze(1)) * torch.sigmoid(eg.unsqueeze(2) + pg.unsqueeze(1))
    out = F.linear(h, fc2.weight, fc2.bias)
    if log_softmax:
        out = F.log_softmax(out if scale == 1.0 else scale * out, dim=-1)
    return out
