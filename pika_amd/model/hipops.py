"""autograd Functions backed by libpika_amd.so (GPU tensors only; nothing here has a CPU path).

Every product -- forward, dX, dW -- runs on the hand-written MFMA kernels behind pika_gemm_nt /
pika_gemm_bf16_*: the direct-to-LDS ping-pong kernels for bf16 operands (plain, time-delay and
reduction-major views), the register-staged kernel for fp32 operands and the exact fp32 parity mode.

In the bf16 arithmetic mode a tensor whose every consumer is an MFMA product is produced in bf16 by the kernel
that computes it (GEMM epilogues, LayerNorm, BatchNorm, the attention core, the joint gate) and its gradient
is accepted in bf16; chains whose intermediates only feed further products are single autograd nodes
(FeedForwardFn, TdnnBnFn, JointOutFn, LinearDropoutResidualFn, PackedAttentionFn) so those intermediates and
their gradients never exist in fp32.  PIKA_NO_FUSED=1 switches all of that off for diagnostics
(tools/mode_diff.py): then only the operand rounding inside pika_gemm_nt separates the mode from fp32.
"""
import ctypes
import os

import torch

from .. import _lib
from .. import gemm as G
from ..rnnt import MAX_FUSED_V


# Module switches (tests flip them; no environment variable): the joint's log-sum-exp partials from the fc2 epilogue, the
# 16-bit logits lattice on top of them, the fused two-term attention of the inference modes.
JOINT_LSE_EPILOGUE = True
JOINT_F16_LOGITS = True
INFER_ATTN = True


def _fused():
    """PIKA_NO_FUSED=1 (diagnostics, tools/mode_diff.py) turns the bf16 fast paths off: every product then goes
    through pika_gemm_nt on fp32 tensors (operands rounded inside the kernel) and torch's attention chain."""
    return os.environ.get("PIKA_NO_FUSED") is None


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _pad4(n):
    return (n + 3) & ~3


def _pad8(n):
    return (n + 7) & ~7


def _tdtype():
    # transposed operands: bf16 halves their traffic; the split modes ("fp32", "bf16x3") need the fp32 bits
    return torch.bfloat16 if G.bf16_backward() else torch.float32


def _mixed():
    """Two-term forward / bf16 backward (pika_amd.gemm.PRECISION == "mixed")."""
    return G.PRECISION == "mixed" and _fused()


class Pair(object):
    """A two-term activation x = hi + lo: two bf16 planes of one buffer (lo directly behind hi), written by the kernel
    that produces the activation and read by the forward product that consumes it (gemm.pair_operand).  `hi` is the
    tensor the bf16 mode would carry -- it is what autograd differentiates and what the backward products read; `lo`
    is a non-differentiable companion.  Only what the model code needs of the tensor interface is provided."""
    __slots__ = ("hi", "lo")

    def __init__(self, hi, lo):
        self.hi, self.lo = hi, lo

    @staticmethod
    def empty(shape, device):
        buf = torch.empty((2,) + tuple(shape), dtype=torch.bfloat16, device=device)
        return Pair(buf[0], buf[1])

    shape = property(lambda self: self.hi.shape)
    dtype = property(lambda self: self.hi.dtype)
    device = property(lambda self: self.hi.device)
    is_cuda = property(lambda self: self.hi.is_cuda)

    def dim(self):
        return self.hi.dim()

    def size(self, *a):
        return self.hi.size(*a)

    def numel(self):
        return self.hi.numel()

    def view(self, *shape):
        return Pair(self.hi.view(*shape), self.lo.view(*shape))

    def reshape(self, *shape):
        return Pair(self.hi.view(*shape), self.lo.view(*shape))     # planes are contiguous: never a copy

    def contiguous(self):
        return self

    def float(self):
        return self.hi.float() + self.lo.float()


def _planes(x, x_lo, K):
    """The two planes (rows, Cp) of the A side of a two-term product and Cp: from a pair that crosses autograd nodes
    (x bf16 + x_lo, K % 64 == 0) or by splitting an f32 tensor here (K % 8 == 0; columns [K, Cp) zero)."""
    if x.dtype == torch.bfloat16:
        assert x_lo is not None and K % 64 == 0 and x.is_contiguous() and x_lo.is_contiguous()
        return x.view(-1, K), x_lo.view(-1, K), K
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or (x2.stride(0) & 3):
        x2 = x2.contiguous()
    hi, lo = G.split_pair(x2.float() if x2.dtype != torch.float32 else x2)
    return hi, lo, hi.shape[1]


def transpose_cast(op, rows, K, device):
    """(K, pad4(rows)) transposed copy of a (virtual) operand, zero-padded columns."""
    ld = _pad8(rows)  # a multiple of 8 so the copy can be a bf16 GEMM operand (16-byte loads)
    dt = _tdtype()
    out = torch.empty((K, ld), dtype=dt, device=device)
    rc = _lib.lib().pika_transpose_cast(ctypes.byref(op), rows, K, out.data_ptr(), ld,
                                        G.PIKA_F32 if dt == torch.float32 else G.PIKA_BF16, _stream())
    _lib.check(rc, "pika_transpose_cast")
    return out


def weight_taps_transposed(w2d, taps=1):
    """(C, taps * N) bf16 operand of a layer's d(input) product from its (N, taps * C) f32 weight: taps reversed and W^T per tap
    (taps = 1: W^T), one launch (include/pika_ops.h) -- what `w.view(N, taps, C).flip(1).permute(2, 1, 0).reshape(C, -1)
    .to(bfloat16)` computes in three."""
    w = w2d.detach()
    if not (w.is_contiguous() and w.dtype == torch.float32):
        w = w.contiguous().float()
    N, K = w.shape
    C = K // taps
    out = torch.empty((C, taps * N), dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.lib().pika_weight_taps_transposed_bf16(w.data_ptr(), N, taps, C, out.data_ptr(), _stream()),
               "pika_weight_taps_transposed_bf16")
    return out


def _colsum_into(out, ptr, ld, rows, cols, device, bf16):
    """Column sums of a (rows, cols) matrix at `ptr` with pitch `ld` into `out`: chunk sums side by side in a scratch tensor,
    folded by a second launch (include/pika_ops.h: no atomics, the same bits every time)."""
    lib = _lib.lib()
    part = torch.empty(int(lib.pika_colsum_partial_floats(rows, cols)), dtype=torch.float32, device=device)
    fn = lib.pika_colsum_bf16 if bf16 else lib.pika_colsum
    _lib.check(fn(ptr, ld, rows, cols, out.data_ptr(), part.data_ptr(), _stream()), "pika_colsum_bf16" if bf16 else "pika_colsum")
    return out


def colsum(x2d):
    out = torch.empty(x2d.shape[1], dtype=torch.float32, device=x2d.device)
    return _colsum_into(out, x2d.data_ptr(), x2d.stride(0), x2d.shape[0], x2d.shape[1], x2d.device, False)


def _bf16_operand(t, min_elems=1 << 21):
    """bf16 copy of a large fp32 activation matrix in the bf16 arithmetic mode.  The MFMA kernel rounds
    fp32 operands to bf16 on their way into LDS anyway, so results are bit-identical; rounding ONCE
    here halves the bytes every GEMM that consumes the matrix pulls through L2 (the forward product
    and the weight-gradient product read the same copy, the three taps of a time-delay operand
    re-read it)."""
    if (G.bf16_backward() and _fused() and t.dtype == torch.float32 and t.numel() >= min_elems
            and t.shape[-1] % 8 == 0):
        return t.to(torch.bfloat16)
    return t


def _weight_for(x, w2d):
    """The weight in the dtype of its activation operand: a bf16 x bf16 product (K % 64 == 0) runs on the
    direct-to-LDS kernel (gemm_glds.hip); the bf16 rounding of W is the one the MFMA path applies anyway."""
    if x.dtype == torch.bfloat16 and w2d.dtype == torch.float32 and w2d.shape[1] % 64 == 0:
        return w2d.detach().to(torch.bfloat16)
    return w2d


def _weight_t(w2d):
    """W^T (K,N) for dX = dY @ W.  Weights are small next to activations; one pass per call."""
    op, rows, K = G.matrix(w2d)
    return transpose_cast(op, rows, K, w2d.device)


def _g(t):
    return 8 if t.dtype == torch.bfloat16 else 4


def _grad_weight(dy2, a_op, a_g, M, Ka, N):
    """dW[N,Ka] = dY^T[N,M] @ A[M,Ka].  Both operands are read as they lie in memory (reduction =
    the row index): `trans` operands + the LDS transpose read, no transposed copies.  Falls back
    to explicit transposes when an output extent is not a multiple of the 16-byte load width."""
    if N % _g(dy2) == 0 and Ka % a_g == 0:
        dy_op = G.matrix(dy2)[0]
        dy_op.trans = 1
        a_op.trans = 1
        out = torch.empty((N, Ka), dtype=torch.float32, device=dy2.device)
        return G.launch(dy_op, a_op, out, Ka, N, Ka, M)
    a_op.trans = 0
    dyt = transpose_cast(G.matrix(dy2)[0], M, N, dy2.device)       # (N, Mp)
    at = transpose_cast(a_op, M, Ka, dy2.device)                    # (Ka, Mp)
    return G.gemm_nt(dyt, at)


def _grad_input(dy2, w2d):
    """dX[M,K] = dY[M,N] @ W[N,K].  W is small next to the activations: a transposed bf16 copy
    (one pass over the weight) feeds the plain NT kernel, which measured faster here than reading
    W in place through a `trans` operand (417 vs 338 us per encoder layer)."""
    M, N = dy2.shape
    K = w2d.shape[1]
    if N % 4 == 0:
        return G.gemm_nt(dy2, _weight_t(w2d))
    Np = _pad4(N)  # reduction length not a multiple of 4: pad the (small) N axis
    dyp = torch.zeros((M, Np), device=dy2.device)
    dyp[:, :N] = dy2
    wt = torch.zeros((K, Np), device=dy2.device)
    wt[:, :N] = w2d.t()
    return G.gemm_nt(dyp, wt)


def _pp_ok(M, N, K):
    """Shapes the fused bf16 epilogues (pika_gemm_bf16_epilogue) accept."""
    return K % 64 == 0 and N % 4 == 0 and M >= 1


def colsum_any(x2d):
    if x2d.dtype == torch.float32:
        return colsum(x2d)
    out = torch.empty(x2d.shape[1], dtype=torch.float32, device=x2d.device)
    return _colsum_into(out, x2d.data_ptr(), x2d.stride(0), x2d.shape[0], x2d.shape[1], x2d.device, True)


class LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b) over the last dim (nn.Linear semantics), optional fused ReLU.

    bf16 activations: x may be a bf16 tensor (the output of an op whose only consumers are MFMA
    products: LayerNorm, the attention core) -- then dx comes back bf16 straight from the GEMM epilogue;
    out_bf16 asks for a bf16 y under the same contract (its gradient arrives bf16, no casts anywhere)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, out_bf16=False, x_lo=None, out_pair=False):
        K = x.shape[-1]
        ctx.set_materialize_grads(False)
        ctx.relu = relu
        ctx.has_bias = bias is not None
        ctx.x_bf16 = x.dtype == torch.bfloat16
        if _mixed() and K % 8 == 0 and weight.shape[0] % 4 == 0 and (x_lo is not None or x.dtype == torch.float32):
            # two-term forward product: hi.hi + lo.hi + hi.lo over a three times longer reduction (gemm.pair_operand)
            N = weight.shape[0]
            with torch.cuda.device(x.device):
                hi, lo, Cp = _planes(x, x_lo, K)
                M = hi.shape[0]
                a_op = G.pair_operand(hi, lo, M, Cp)
                wb = G.split_weight(weight)
                if out_pair:
                    out = Pair.empty(x.shape[:-1] + (N,), x.device)
                    G.gemm_ex(a_op, wb, M, N, 3 * Cp, G.EPI_DROPOUT_BF16, out.hi.view(-1, N), out_lo=out.lo.view(-1, N),
                              bias=bias, relu=relu)
                else:
                    out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
                    G.gemm_ex(a_op, wb, M, N, 3 * Cp, G.EPI_F32, out.view(-1, N), bias=bias, relu=relu)
            y = out.hi if out_pair else out
            ctx.save_for_backward(hi[:, :K], weight, y if relu == 1 else None)   # the backward reads the hi plane
            if out_pair:
                ctx.mark_non_differentiable(out.lo)
                return out.hi, out.lo
            return out
        if x_lo is not None:        # a two-plane input of a product the two-term kernel does not take (output pitch)
            x = x.float() + x_lo.float()
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or (x2.stride(0) & 3):
            x2 = x2.contiguous()
        x2 = _bf16_operand(x2)
        N = weight.shape[0]
        M = x2.shape[0]
        with torch.cuda.device(x.device):
            if out_bf16 and x2.dtype == torch.bfloat16 and _pp_ok(M, N, K):
                out = torch.empty(x.shape[:-1] + (N,), dtype=torch.bfloat16, device=x.device)
                _gemm_epilogue(x2, weight.detach().to(torch.bfloat16), out.view(-1, N), bias, EPI_DROPOUT_BF16,
                               relu=1 if relu else 0)
            else:
                out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)  # not a view:
                G.gemm_nt(x2, _weight_for(x2, weight), bias=bias, relu=relu, out=out.view(-1, N),  # may be
                          precision="fp32" if (_mixed() and x2.dtype == torch.float32) else None)
                if out_bf16:                                                        # overwritten in place
                    out = out.to(torch.bfloat16)
        ctx.save_for_backward(x2, weight, out if relu == 1 else None)
        return out

    @staticmethod
    def backward(ctx, dy, *_):
        x2, weight, y = ctx.saved_tensors
        N, K = weight.shape
        dy2 = dy.reshape(-1, N)
        if ctx.relu == 1:   # relu == 2: the mask was already applied by the consumer's backward
            dy2 = dy2 * (y.view(-1, N) > 0)
        dy2 = dy2.contiguous()
        M = dy2.shape[0]
        dx = dw = db = None
        with torch.cuda.device(dy.device):
            dyb = _bf16_operand(dy2)
            if ctx.needs_input_grad[0]:
                if ctx.x_bf16 and dyb.dtype == torch.bfloat16 and N % 64 == 0 and K % 4 == 0:
                    dx = torch.empty(dy.shape[:-1] + (K,), dtype=torch.bfloat16, device=dy.device)
                    _gemm_epilogue(dyb, _weight_t(weight), dx.view(-1, K), None, EPI_DROPOUT_BF16)
                else:
                    dx = _grad_input(dyb, weight).view(*dy.shape[:-1], K)
                    if ctx.x_bf16:
                        dx = dx.to(torch.bfloat16)
            if ctx.needs_input_grad[1]:
                dw = _grad_weight(dyb, G.matrix(x2)[0], _g(x2), M, K, N)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = colsum_any(dy2)
        return dx, dw, db, None, None, None, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim (include/pika_norm.h); out_bf16: the result only feeds MFMA
    products, so it is produced (and its gradient accepted) in bf16."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_bf16, out_pair=False, with_skip=False):
        """with_skip: x itself is returned as the LAST output (autograd makes it an alias whose gradient arrives in this
        node's backward): the caller's skip connection around the LN then costs no gradient-accumulation launch -- the
        backward kernel adds the skip's gradient to dx on the way out."""
        ctx.with_skip = bool(with_skip)
        C = x.shape[-1]
        x2 = x.reshape(-1, C).contiguous()
        rows = x2.shape[0]
        ctx.set_materialize_grads(False)
        out_bf16 = out_bf16 or out_pair
        dt = torch.bfloat16 if out_bf16 else torch.float32
        pair = Pair.empty(x.shape, x.device) if out_pair else None      # two planes: y = hi + lo
        y = pair.hi if out_pair else torch.empty(x.shape, dtype=dt, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().pika_layer_norm_fwd(
                x2.data_ptr(), rows, C, weight.data_ptr(), bias.data_ptr(), float(eps), y.data_ptr(),
                G.PIKA_BF16 if out_bf16 else G.PIKA_F32, pair.lo.data_ptr() if out_pair else None, mean.data_ptr(),
                rstd.data_ptr(), _stream()), "pika_layer_norm_fwd")
        ctx.save_for_backward(x2, weight, mean, rstd)
        if out_pair:
            ctx.mark_non_differentiable(pair.lo)
            return (pair.hi, pair.lo, x) if with_skip else (pair.hi, pair.lo)
        return (y, x) if with_skip else y

    @staticmethod
    def backward(ctx, dy, *rest):
        x2, weight, mean, rstd = ctx.saved_tensors
        rows, C = x2.shape
        dskip = rest[-1] if (ctx.with_skip and rest) else None
        if dy is None:                  # only the skip connection reached the loss
            return dskip, None, None, None, None, None, None
        if dskip is not None:
            dskip = dskip.reshape(rows, C)
            if dskip.dtype != torch.float32 or not dskip.is_contiguous():
                dskip = dskip.float().contiguous()
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        dy = dy.contiguous()
        dx = torch.empty(dy.shape, dtype=torch.float32, device=dy.device)
        dg = torch.empty(C, dtype=torch.float32, device=dy.device)
        db = torch.empty(C, dtype=torch.float32, device=dy.device)
        # the workgroups' column sums side by side, added up by a second launch in a fixed order (no atomics, no memsets)
        part = torch.empty(int(_lib.lib().pika_layer_norm_bwd_partial_floats(rows, C)), dtype=torch.float32, device=dy.device)
        with torch.cuda.device(dy.device):
            _lib.check(_lib.lib().pika_layer_norm_bwd(
                dy.data_ptr(), G.PIKA_F32 if dy.dtype == torch.float32 else G.PIKA_BF16, x2.data_ptr(), rows, C,
                weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                part.data_ptr(), None if dskip is None else dskip.data_ptr(), _stream()), "pika_layer_norm_bwd")
        return dx, dg, db, None, None, None, None


class TimeDelayFn(torch.autograd.Function):
    """y[b,t,n] = act(sum_{tap,c} W[n,tap*C+c] * x[b, t*stride + tap*dil - pad, c] + bias[n]).

    TDNN layers (pad 0) and the causal Conv1d of the prediction net (taps 5, pad 4) as ONE GEMM
    over a virtual operand: im2col is never materialised in the forward pass."""

    @staticmethod
    def forward(ctx, x, w2d, bias, taps, dil, stride, pad, relu):
        Bn, T, C = x.shape
        N = w2d.shape[0]
        if _mixed() and x.dtype == torch.float32 and C % 8 == 0:
            # two-term forward product over a time-delay view of the two planes (3 * Cp reduction columns per tap)
            with torch.cuda.device(x.device):
                hi, lo, Cp = _planes(x.contiguous(), None, C)
                t_out = G.time_delay(x, taps, dil, stride, pad)[3]
                a_op = G.pair_operand(hi, lo, Bn * t_out, Cp, taps, dil, stride, pad, rows_per_batch=t_out, t_in=T,
                                      batch_stride=T * Cp)
                y = torch.empty((Bn, t_out, N), dtype=torch.float32, device=x.device)
                G.gemm_ex(a_op, G.split_weight(w2d, taps), Bn * t_out, N, taps * 3 * Cp, G.EPI_F32, y.view(-1, N), bias=bias,
                          relu=relu)
            ctx.cfg = (taps, dil, stride, pad, relu, t_out)
            ctx.has_bias = bias is not None
            ctx.save_for_backward(hi.view(Bn, T, Cp)[:, :, :C], w2d, y if relu == 1 else None)
            return y
        x = _bf16_operand(x.contiguous())
        with torch.cuda.device(x.device):
            a_op, M, K, t_out = G.time_delay(x, taps, dil, stride, pad)
            y = torch.empty((Bn, t_out, N), dtype=torch.float32, device=x.device)
            G.launch(a_op, G.matrix(_weight_for(x, w2d))[0], y, N, M, N, K, bias=bias, relu=relu,
                     precision="fp32" if (_mixed() and x.dtype == torch.float32) else None)
        ctx.cfg = (taps, dil, stride, pad, relu, t_out)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w2d, y if relu == 1 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2d, y = ctx.saved_tensors
        taps, dil, stride, pad, relu, t_out = ctx.cfg
        Bn, T, C = x.shape
        N, K = w2d.shape
        dy2 = dy.reshape(-1, N)
        if relu == 1:
            dy2 = dy2 * (y.view(-1, N) > 0)
        dy2 = dy2.contiguous()
        M = dy2.shape[0]
        dx = dw = db = None
        with torch.cuda.device(dy.device):
            dyb = _bf16_operand(dy2)
            if ctx.needs_input_grad[0] and stride == 1 and dyb.dtype == torch.bfloat16 and N % 64 == 0:
                # transposed convolution as ONE GEMM over a padded time-delay view of dY (taps reversed):
                # dx[b,ti,c] = sum_{tap',n} dY[b, ti + tap'*dil - pad', n] * W[n, (taps-1-tap')*C + c],
                # pad' = (taps-1)*dil - pad; rows outside [0,t_out) read as zeros.  No (M, taps*C) column
                # gradient, no col2im pass.
                wrev = weight_taps_transposed(w2d, taps)
                a_op = G.Operand(dyb.data_ptr(), G.PIKA_BF16, T, t_out, t_out * N, N, N, 1, dil,
                                 (taps - 1) * dil - pad, 0, 0)
                dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
                G.launch(a_op, G.matrix(wrev)[0], dx.view(-1, C), C, Bn * T, C, taps * N)
            elif ctx.needs_input_grad[0]:
                dcol = _grad_input(dyb, w2d)  # (M, taps*C)
                dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
                _lib.check(_lib.lib().pika_col2im(dcol.data_ptr(), dx.data_ptr(), Bn, t_out, T, C,
                                                  taps, stride, dil, pad, _stream()), "pika_col2im")
            if ctx.needs_input_grad[1]:
                a_op = G.time_delay(x, taps, dil, stride, pad)[0]
                dw = _grad_weight(dyb, a_op, _g(x), M, K, N)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = colsum(dy2)
        return dx, dw, db, None, None, None, None, None


class GateFn(torch.autograd.Function):
    """h[b,t,u,:] = tanh(e1[b,t]+p1[b,u]) * sigmoid(eg[b,t]+pg[b,u])  (include/pika_joint.h).

    Output is bf16 in the bf16 arithmetic mode (it only feeds the fc2 MFMA GEMM, which would
    round it anyway) and fp32 in the parity mode.  Nothing of size B*T*U*H is saved: the backward
    recomputes tanh/sigmoid from the four small inputs."""

    @staticmethod
    def forward(ctx, e1, p1, eg, pg):
        e1, p1, eg, pg = [t.contiguous() for t in (e1, p1, eg, pg)]
        Bn, T, H = e1.shape
        U = p1.shape[1]
        dt = torch.bfloat16 if G.joint_in_bf16() else torch.float32
        h = torch.empty((Bn, T, U, H), dtype=dt, device=e1.device)
        with torch.cuda.device(e1.device):
            _lib.check(_lib.lib().pika_joint_gate_fwd(
                e1.data_ptr(), p1.data_ptr(), eg.data_ptr(), pg.data_ptr(), h.data_ptr(),
                G.PIKA_F32 if dt == torch.float32 else G.PIKA_BF16, Bn, T, U, H, _stream()),
                "pika_joint_gate_fwd")
        ctx.save_for_backward(e1, p1, eg, pg)
        return h

    @staticmethod
    def backward(ctx, dh):
        e1, p1, eg, pg = ctx.saved_tensors
        Bn, T, H = e1.shape
        U = p1.shape[1]
        if dh.dtype not in (torch.float32, torch.bfloat16):
            dh = dh.float()
        dh = dh.contiguous()
        de1, deg = torch.empty_like(e1), torch.empty_like(eg)
        dp1, dpg = torch.empty_like(p1), torch.empty_like(pg)
        with torch.cuda.device(dh.device):
            _lib.check(_lib.lib().pika_joint_gate_bwd(
                dh.data_ptr(), G.PIKA_F32 if dh.dtype == torch.float32 else G.PIKA_BF16, e1.data_ptr(), p1.data_ptr(), eg.data_ptr(), pg.data_ptr(),
                de1.data_ptr(), dp1.data_ptr(), deg.data_ptr(), dpg.data_ptr(), Bn, T, U, H,
                _stream()), "pika_joint_gate_bwd")
        return de1, dp1, deg, dpg


class LogSoftmaxFn(torch.autograd.Function):
    """log_softmax(scale * x) over the last dim, IN PLACE on x (the (B,T,U,V) logits buffer is
    7.8 GB at config 2); backward rewrites the incoming dense gradient in place as well."""

    @staticmethod
    def forward(ctx, x, scale):
        assert x.is_contiguous()
        cols = x.shape[-1]
        rows = x.numel() // cols
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().pika_log_softmax_rows(x.data_ptr(), rows, cols, cols, float(scale),
                                                        _stream()), "pika_log_softmax_rows")
        ctx.mark_dirty(x)
        ctx.scale = float(scale)
        ctx.save_for_backward(x)
        return x

    @staticmethod
    def backward(ctx, g):
        (lp,) = ctx.saved_tensors
        cols = lp.shape[-1]
        rows = lp.numel() // cols
        if not g.is_contiguous():
            g = g.contiguous()
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().pika_log_softmax_bwd_rows(lp.data_ptr(), g.data_ptr(), rows, cols,
                                                            cols, ctx.scale, _stream()),
                       "pika_log_softmax_bwd_rows")
        return g, None


def joint_out_ok(h, weight, units=None):
    """(units: the size of the output layer when it is not weight.shape[0] -- a layer about to be padded.)
    JointOutFn preconditions: bf16 hidden, reduction a multiple of 64, vocabulary a multiple of
    4 (8-byte bf16 groups of the d(logits) copy; the 16-bit logits of a vocabulary that is not a multiple of 8 -- the
    shipped recipes' 6268 -- get a pitch of whole 16-byte granules) that one wave covers (log-softmax row kernels)."""
    N, K = weight.shape
    N = N if units is None else int(units)
    # N <= 8192: what the d(logits) kernels of the backward take (one wave holds a 64-padded row: 64 x 4 x 32 columns)
    return (G.joint_in_bf16() and _fused() and h.dtype == torch.bfloat16 and K % 64 == 0 and N % 4 == 0
            and N <= MAX_FUSED_V)


class JointOutFn(torch.autograd.Function):
    """log_softmax(scale * (h @ W2^T + b2)) over the (B,T,U,V) lattice: the 8 GFLOP-per-utterance
    tail of the joint network (reference trainer/model/transducer.py:107-111), bf16 arithmetic mode.

    forward : pika_gemm_bf16_nt (h bf16, bf16 copy of W2) -> logits f32, log-softmax in place.
    backward: the dense RNN-T gradient g and the saved log-probs are folded into ONE bf16 matrix
              d(logits) (M, pad64(V)) -- 2 bytes/element written instead of 4 rewritten in place,
              and every consumer reads half the bytes: dh = d(logits) @ W2 (direct-to-LDS GEMM,
              zero-padded reduction), dW2 = d(logits)^T @ h (both read in place as `trans`
              operands), db2 = column sums."""

    compact_hits = 0   # times the backward used the loss' compact gradient (tests / diagnostics)

    @staticmethod
    def forward(ctx, h, weight, bias, scale, lazy=False, labels=None, width=None):
        # width (lazy output only): the caller's vocabulary when `weight` / `bias` are its output layer padded to a
        # multiple of four units (pika_amd.model.ops.joint): the lazy tensor returned stands for `width` columns
        N, K = weight.shape
        h2 = h.reshape(-1, K)
        ctx.scale = float(scale)
        ctx.has_bias = bias is not None
        ctx.state = None
        ctx.width = None if width is None or int(width) == N else int(width)
        assert ctx.width is None or (lazy and h.dim() == 4), "a padded output layer needs the lazy output"
        # (N <= 8192: what the loss' fused path and the compact d(logits) kernels take; beyond, the fp16 copy would only be
        # followed by a second, fp32 run of the product for the loss -- ADVICE r4)
        if (lazy and h.dim() == 4 and labels is not None and scale == 1.0 and 256 < N <= MAX_FUSED_V and N % 4 == 0
                and labels.dim() == 2 and labels.shape == (h.shape[0], h.shape[2] - 1)
                and JOINT_LSE_EPILOGUE and JOINT_F16_LOGITS):
            # 16-bit logits: the (B,T,U1,V) lattice -- the largest tensor of a training step -- is written ONCE as fp16
            # and read once (by the d(logits) pass of the backward); everything the LOSS reads leaves the product's
            # epilogue in fp32: the row log-sum-exp partials and the logits of the blank and of the row's label, so costs
            # are those of fp32 logits.  Any other reader gets log-probabilities from a re-run of the product in fp32
            # (LogitsState.recompute): values never come from the fp16 copy except in the backward's softmax, whose result
            # is rounded to bf16 anyway (include/pika_gemm.h: pika_gemm_bf16_nt_lse_f16)
            from ..rnnt import LazyLogProbs, LogitsState
            B_, T_, U1_ = h.shape[0], h.shape[1], h.shape[2]
            M = h2.shape[0]
            Nld = (N + 7) & ~7          # rows of whole 16-byte granules
            out16 = torch.empty(h.shape[:-1] + (Nld,), dtype=torch.float16, device=h.device)
            if Nld != N:
                out16 = out16[..., :N]
            n_part = (N + 255) // 256 * 4
            part = torch.empty((2, M, n_part), dtype=torch.float32, device=h.device)
            gath = torch.empty((M, 2), dtype=torch.float32, device=h.device)
            lab32 = labels.detach().to(torch.int32).contiguous()
            wb = weight.detach().to(torch.bfloat16)
            with torch.cuda.device(h.device):
                _lib.check(_lib.lib().pika_gemm_bf16_nt_lse_f16(
                    h2.data_ptr(), h2.stride(0), wb.data_ptr(), wb.stride(0), out16.data_ptr(), Nld, M, N, K,
                    None if bias is None else bias.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), n_part,
                    lab32.data_ptr(), T_, U1_, 0, gath.data_ptr(), _stream()), "pika_gemm_bf16_nt_lse_f16")
            st = ctx.state = LogitsState(scale)
            st.partials, st.gathered = part, (gath, lab32, 0)
            h2d, wd, bd = h2.detach(), weight.detach(), None if bias is None else bias.detach()

            def recompute():
                full = torch.empty((B_, T_, U1_, N), dtype=torch.float32, device=h2d.device)
                G.gemm_bf16_nt(h2d, wd.to(torch.bfloat16), bias=bd, out=full.view(-1, N))
                with torch.cuda.device(h2d.device):
                    _lib.check(_lib.lib().pika_log_softmax_rows(full.data_ptr(), M, N, N, 1.0, _stream()),
                               "pika_log_softmax_rows")
                return full
            st.recompute = recompute
            ctx.save_for_backward(h2, weight, out16)
            return LazyLogProbs(st, out16, ctx.width)
        out = torch.empty(h.shape[:-1] + (N,), dtype=torch.float32, device=h.device)
        if lazy and out.dim() == 4:
            # the log-softmax pass is deferred until something needs the values (pika_amd.rnnt.LazyLogProbs);
            # `out` is normalised in place by a raw kernel call, which autograd's version counter does not see.
            # The GEMM's epilogue leaves per-row partial (max, sum exp) pairs per 64-column block, from which this
            # package's loss takes the row log-sum-exp without reading the logits again (include/pika_gemm.h)
            from ..rnnt import LazyLogProbs, LogitsState
            ctx.state = LogitsState(scale)
            M = h2.shape[0]
            if scale == 1.0 and N > 256 and JOINT_LSE_EPILOGUE:
                n_part = (N + 255) // 256 * 4
                part = torch.empty((2, M, n_part), dtype=torch.float32, device=h.device)
                wb = weight.detach().to(torch.bfloat16)
                with torch.cuda.device(h.device):
                    _lib.check(_lib.lib().pika_gemm_bf16_nt_lse(
                        h2.data_ptr(), h2.stride(0), wb.data_ptr(), wb.stride(0), out.data_ptr(), N, M, N, K,
                        None if bias is None else bias.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), n_part,
                        _stream()), "pika_gemm_bf16_nt_lse")
                ctx.state.partials = part
            else:
                G.gemm_bf16_nt(h2, weight.detach().to(torch.bfloat16), bias=bias, out=out.view(-1, N))
            ctx.save_for_backward(h2, weight, out)
            return LazyLogProbs(ctx.state, out, ctx.width)
        G.gemm_bf16_nt(h2, weight.detach().to(torch.bfloat16), bias=bias, out=out.view(-1, N))
        ctx.save_for_backward(h2, weight, out)
        with torch.cuda.device(h.device):
            _lib.check(_lib.lib().pika_log_softmax_rows(out.data_ptr(), h2.shape[0], N, N, float(scale),
                                                        _stream()), "pika_log_softmax_rows")
        return out

    @staticmethod
    def backward(ctx, g):
        h2, weight, lp = ctx.saved_tensors
        N, K = weight.shape
        M = h2.shape[0]
        Np = (N + 63) & ~63
        from ..rnnt import LazyDenseGrad
        compact = None
        lse = None
        if isinstance(g, LazyDenseGrad):
            # the loss' own gradient, never written: its non-zeros are in the loss workspace.  (Once something has
            # made it dense, or when it does not fit the compact kernel, it is an ordinary tensor from here on.)
            # (the workspace is laid out for the columns the KERNELS saw: `lp`'s, also when the caller sees fewer -- ctx.width)
            if g._dense is None and lp.dim() == 4 and tuple(lp.shape) == tuple(g.compact.dims[:4]) and N <= MAX_FUSED_V:
                compact = g.compact
                if ctx.state is not None and ctx.state.raw:
                    lse = g.lse     # the loss read the raw logits `lp` still holds: log-prob = logit - lse
            else:
                g = g.dense()
        if ctx.width is not None and compact is None:
            # a dense gradient over the caller's columns: zeros for the padding units of the output layer
            g = torch.nn.functional.pad(g.dense() if isinstance(g, LazyDenseGrad) else g, (0, N - ctx.width))
        lp16 = None
        if lp.dtype == torch.float16:      # 16-bit logits: only the compact gradient on the raw logits reads them ...
            if compact is not None and lse is not None:
                lp16 = lp
            else:                          # ... anything else works on fp32 log-probabilities (the product runs again)
                lp = ctx.state.to_log_probs(lp)
                lse = None
        elif lse is None and ctx.state is not None:
            ctx.state.to_log_probs(lp)     # every other path below needs the log-probabilities in `lp`
        if compact is None:
            if not g.is_contiguous():
                g = g.contiguous()
            c = getattr(g, "_pika_compact", None)
            if c is not None and c.matches(g) and lp.dim() == 4 and tuple(lp.shape) == tuple(g.shape) and N <= MAX_FUSED_V:
                compact = c
        dh = dw = db = None
        with torch.cuda.device(lp.device):
            dl = torch.empty((M, Np), dtype=torch.bfloat16, device=lp.device)
            db_fused = None
            if compact is not None:
                # g is the RNN-T loss' own dense gradient, untouched: take its two non-zeros per row from the
                # loss workspace instead of reading 4 bytes x B*T*U*V back (include/pika_rnnt.h)
                B_, T_, U1_, V_, blank = compact.dims
                want_db = ctx.has_bias and ctx.needs_input_grad[2]
                db_fused = torch.empty(N, dtype=torch.float32, device=dl.device) if want_db else None
                if lp16 is not None:
                    # the blank / label logits of every row in fp32, where the forward product kept them (the fp16
                    # copy is 2^-11 of |logit| off -- percents of a softmax value -- on the two entries that matter most)
                    gath = ctx.state.gathered if ctx.state is not None else None
                    _lib.check(_lib.lib().pika_rnnt_dlogits_compact_bf16_f16in(
                        lp16.data_ptr(), lp16.stride(-2), lse.data_ptr(), compact.ws.data_ptr(), B_, T_, U1_, V_,
                        blank, dl.data_ptr(), Np, ctx.scale,
                        None if db_fused is None else db_fused.data_ptr(),
                        None if gath is None else gath[0].data_ptr(), None if gath is None else gath[1].data_ptr(),
                        0 if gath is None else int(gath[2]), _stream()), "pika_rnnt_dlogits_compact_bf16_f16in")
                else:
                    _lib.check(_lib.lib().pika_rnnt_dlogits_compact_bf16(
                        lp.data_ptr(), None if lse is None else lse.data_ptr(), compact.ws.data_ptr(), B_, T_, U1_, V_,
                        blank, dl.data_ptr(), Np, ctx.scale,
                        None if db_fused is None else db_fused.data_ptr(), _stream()), "pika_rnnt_dlogits_compact_bf16")
                JointOutFn.compact_hits += 1
            else:
                _lib.check(_lib.lib().pika_log_softmax_bwd_rows_bf16(
                    lp.data_ptr(), g.data_ptr(), dl.data_ptr(), M, N, N, Np, ctx.scale, _stream()),
                    "pika_log_softmax_bwd_rows_bf16")
            del g
            if ctx.needs_input_grad[0]:
                wt = torch.zeros((K, Np), dtype=torch.bfloat16, device=dl.device)
                wt[:, :N] = weight.detach().t()
                # bf16 like h itself: autograd would round an fp32 dh to h's dtype anyway (two extra passes)
                dh = torch.empty(lp.shape[:-1] + (K,), dtype=torch.bfloat16, device=dl.device)
                _gemm_epilogue(dl, wt, dh.view(-1, K), None, EPI_DROPOUT_BF16)
            if ctx.needs_input_grad[1]:
                # (N % 8 != 0: over the padded columns of d(logits) -- zeros -- and the rows cut off: the in-place `trans`
                # operands want whole 16-byte granules)
                dw = _grad_weight(dl[:, :N], G.matrix(h2)[0], 8, M, K, N) if N % 8 == 0 else \
                    _grad_weight(dl, G.matrix(h2)[0], 8, M, K, Np)[:N]
            if db_fused is not None:
                db = db_fused           # column sums came out of the d(logits) kernel itself
            elif ctx.has_bias and ctx.needs_input_grad[2]:
                db = torch.empty(N, dtype=torch.float32, device=dl.device)
                _colsum_into(db, dl.data_ptr(), Np, M, N, dl.device, True)
        return dh, dw, db, None, None, None, None


def attention_ok(q, k, v, heads, mask):
    """AttentionFn preconditions: the encoder's self-attention (no mask, Tq == Tk), head width 64
    or 128, bf16 arithmetic mode."""
    D = q.shape[-1] // heads
    if mask is not None and (mask.dim() != 3 or mask.shape[1] != q.shape[1] or mask.shape[2] != k.shape[1]):
        return False
    return (G.bf16_backward() and _fused() and q.is_cuda and q.dtype in (torch.float32, torch.bfloat16)
            and q.shape == k.shape == v.shape and D in (64, 128) and D * heads == q.shape[-1])


def attention_keep_mask(BH, T, p_drop, seed, device):
    """(BH,T,T) bool keep-mask the fused kernels use for (p_drop, seed) -- for tests."""
    m = torch.empty((BH, T, T), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().pika_attention_keep_mask(m.data_ptr(), BH, T, float(p_drop), int(seed), _stream()),
                   "pika_attention_keep_mask")
    return m.bool()


def _mask_bytes(mask):
    """The packed form the attention kernels read (include/pika_attn.h: u64 [B][T][ceil(T/64)], bit set = masked) of a
    boolean (B, T, T) attention mask, or None."""
    if mask is None:
        return None
    B, T, Tk = mask.shape
    assert T == Tk
    m8 = mask.to(torch.uint8).contiguous()
    bits = torch.empty((B, T, (T + 63) // 64), dtype=torch.int64, device=mask.device)
    with torch.cuda.device(mask.device):
        _lib.check(_lib.lib().pika_attention_mask_bits(m8.data_ptr(), B, T, bits.data_ptr(), _stream()),
                   "pika_attention_mask_bits")
    return bits


def _attn_fwd(q, k, v, out, lse, B, T, heads, D, ld, p_drop, seed, mask=None, lo_off=None, out_lo_off=0):
    """Returns the packed keep bits (None without dropout) the backward must be given.  lo_off: q, k, v (and out) are the
    hi planes of two-term tensors whose lo planes lie lo_off (out_lo_off) elements behind: two-term forward."""
    bits = None
    if p_drop > 0:
        bits = torch.empty((B * heads, T, (T + 63) // 64), dtype=torch.int64, device=out.device)
    with torch.cuda.device(out.device):
        if lo_off is not None:
            _lib.check(_lib.lib().pika_attention_fwd_two_term(
                q.data_ptr(), k.data_ptr(), v.data_ptr(), lo_off, out.data_ptr(), out_lo_off, lse.data_ptr(),
                None if bits is None else bits.data_ptr(), None if mask is None else mask.data_ptr(), B, T, heads, D, ld,
                heads * D, float(p_drop), int(seed), _stream()), "pika_attention_fwd_two_term")
        else:
            _lib.check(_lib.lib().pika_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
                                                     G.PIKA_F32 if q.dtype == torch.float32 else G.PIKA_BF16,
                                                     lse.data_ptr(), None if bits is None else bits.data_ptr(),
                                                     None if mask is None else mask.data_ptr(),
                                                     B, T, heads, D, ld, heads * D, float(p_drop),
                                                     int(seed), _stream()), "pika_attention_fwd")
    return bits


def _attn_bwd(q, k, v, out, dout, lse, bits, dq, dk, dv, B, T, heads, D, ld, p_drop, seed, mask=None):
    delta = torch.empty_like(lse)
    with torch.cuda.device(out.device):
        _lib.check(_lib.lib().pika_attention_bwd(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(),
            G.PIKA_F32 if q.dtype == torch.float32 else G.PIKA_BF16, lse.data_ptr(),
            None if bits is None else bits.data_ptr(), None if mask is None else mask.data_ptr(), delta.data_ptr(),
            dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, T, heads, D, ld, heads * D,
            p_drop, seed, _stream()), "pika_attention_bwd")


def attention_infer_ok(q, k, v, heads, mask):
    """Inference (no autograd) in one of the two-term arithmetic modes -- the decoder's encoder pass
    (decoder/transducer_decoder.py: "fp16x2" by default, "bf16x3" on request; the exact mode "fp32" keeps the exact torch
    chain): the fused two-term attention takes the fp32 projections, so no decode mode materialises the (B,H,T,T) scores
    or touches hipBLASLt.  INFER_ATTN = False keeps the torch chain everywhere."""
    D = q.shape[-1] // heads
    if mask is not None and (mask.dim() != 3 or mask.shape[1] != q.shape[1] or mask.shape[2] != k.shape[1]):
        return False
    return (not torch.is_grad_enabled() and G.PRECISION in ("fp16x2", "bf16x3") and _fused() and q.is_cuda
            and q.dtype == torch.float32 and q.dim() == 3 and q.shape == k.shape == v.shape and D in (64, 128)
            and D * heads == q.shape[-1] and INFER_ATTN)


def attention_infer_two_term(q, k, v, heads, mask=None):
    """softmax(q k^T / sqrt(D)) v on fp32 projections with two 16-bit terms per operand on the MFMA pipe, softmax in
    fp32; the (B,H,T,T) score tensor -- 8 GB at B = 64, T = 994 -- is never materialised (reference
    multi_headed_attn.py:199-231 builds it three times over).
      "fp16x2": x = hi + 2^-11 lo' (fp16 terms, 22 mantissa bits; pika_attention_infer_f16x2): the decoder's default grade;
      "bf16x3": x = hi + lo (bf16 terms, 16 bits; pika_attention_fwd_two_term, the training forward's kernel)."""
    B, T, HD = q.shape
    if G.PRECISION == "fp16x2":
        planes = torch.empty((2, B, T, 3 * HD), dtype=torch.float16, device=q.device)
        hi, lo = planes[0], planes[1]
        for i, x in enumerate((q, k, v)):
            h = hi[..., i * HD:(i + 1) * HD]
            h.copy_(x.clamp(-65504.0, 65504.0))
            lo[..., i * HD:(i + 1) * HD].copy_((x - h.float()) * 2048.0)
        out = torch.empty((B, T, HD), dtype=torch.float32, device=q.device)
        mb = _mask_bytes(mask)
        with torch.cuda.device(q.device):
            _lib.check(_lib.lib().pika_attention_infer_f16x2(
                hi[..., :HD].data_ptr(), hi[..., HD:2 * HD].data_ptr(), hi[..., 2 * HD:].data_ptr(), hi.numel(),
                out.data_ptr(), None if mb is None else mb.data_ptr(), B, T, heads, HD // heads, 3 * HD, HD, _stream()),
                "pika_attention_infer_f16x2")
        return out
    planes = torch.empty((2, B, T, 3 * HD), dtype=torch.bfloat16, device=q.device)
    hi, lo = planes[0], planes[1]
    for i, x in enumerate((q, k, v)):
        h = hi[..., i * HD:(i + 1) * HD]
        h.copy_(x)
        lo[..., i * HD:(i + 1) * HD].copy_(x - h.float())
    out = Pair.empty((B, T, HD), q.device)
    lse = torch.empty(B * heads * T, dtype=torch.float32, device=q.device)
    _attn_fwd(hi[..., :HD], hi[..., HD:2 * HD], hi[..., 2 * HD:], out.hi, lse, B, T, heads, HD // heads, 3 * HD, 0.0, 0,
              _mask_bytes(mask), lo_off=hi.numel(), out_lo_off=out.hi.numel())
    return out.hi.float().add_(out.lo)


def attention_infer_packed(qkv, heads, mask=None):
    """attention_infer_two_term on the PACKED projection [q | k | v] (B,T,3*H*D) of a self-attention layer: the two 16-bit
    planes of all three come out of ONE split launch (pika_split_bf16_terms, PIKA_SPLIT_PAIR) instead of six element-wise
    passes per tensor -- 18 launches, 1.45 ms per layer of the decoder's encoder pass at B = 64, T = 994; same values."""
    B, T, HD3 = qkv.shape
    HD = HD3 // 3
    f16 = G.PRECISION == "fp16x2"
    x2 = qkv.reshape(B * T, HD3)
    planes = torch.empty((2, B, T, HD3), dtype=torch.float16 if f16 else torch.bfloat16, device=qkv.device)
    mb = _mask_bytes(mask)
    with torch.cuda.device(qkv.device):
        _lib.check(_lib.lib().pika_split_bf16_terms(x2.data_ptr(), 1, B * T, HD3, 0, x2.stride(0), 0, 4 if f16 else 2, 2, HD3,
                                                    planes.data_ptr(), _stream()), "pika_split_bf16_terms")
        hi = planes[0]
        if f16:
            out = torch.empty((B, T, HD), dtype=torch.float32, device=qkv.device)
            _lib.check(_lib.lib().pika_attention_infer_f16x2(
                hi[..., :HD].data_ptr(), hi[..., HD:2 * HD].data_ptr(), hi[..., 2 * HD:].data_ptr(), hi.numel(),
                out.data_ptr(), None if mb is None else mb.data_ptr(), B, T, heads, HD // heads, HD3, HD, _stream()),
                "pika_attention_infer_f16x2")
            return out
    out = Pair.empty((B, T, HD), qkv.device)
    lse = torch.empty(B * heads * T, dtype=torch.float32, device=qkv.device)
    _attn_fwd(hi[..., :HD], hi[..., HD:2 * HD], hi[..., 2 * HD:], out.hi, lse, B, T, heads, HD // heads, HD3, 0.0, 0, mb,
              lo_off=hi.numel(), out_lo_off=out.hi.numel())
    return out.hi.float().add_(out.lo)


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(D)) [dropout] v per head on (B,T,H*D) projections
    (multi_headed_attn.py:199-231) without materialising the (B,H,T,T) tensors: include/pika_attn.h."""

    @staticmethod
    def forward(ctx, q, k, v, heads, p_drop, seed, mask=None):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, T, HD = q.shape
        out = torch.empty_like(q)
        lse = torch.empty(B * heads * T, dtype=torch.float32, device=q.device)
        mask = _mask_bytes(mask)
        bits = _attn_fwd(q, k, v, out, lse, B, T, heads, HD // heads, HD, p_drop, seed, mask)
        ctx.cfg = (heads, float(p_drop), int(seed))
        ctx.save_for_backward(q, k, v, out, lse, bits, mask)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, bits, mask = ctx.saved_tensors
        heads, p_drop, seed = ctx.cfg
        B, T, HD = q.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        _attn_bwd(q, k, v, out, dout.contiguous(), lse, bits, dq, dk, dv, B, T, heads, HD // heads, HD, p_drop, seed,
                  mask)
        return dq, dk, dv, None, None, None, None


class PackedAttentionFn(torch.autograd.Function):
    """The same on ONE packed projection qkv (B,T,3*H*D) = [q | k | v] (self-attention: the three
    projections share their input, so they are one GEMM); the gradient comes back packed as well."""

    @staticmethod
    def forward(ctx, qkv, heads, p_drop, seed, qkv_lo=None, mask=None):
        qkv = qkv.contiguous()
        B, T, HD3 = qkv.shape
        HD = HD3 // 3
        ctx.set_materialize_grads(False)
        q, k, v = qkv[..., :HD], qkv[..., HD:2 * HD], qkv[..., 2 * HD:]
        lse = torch.empty(B * heads * T, dtype=torch.float32, device=qkv.device)
        mask = _mask_bytes(mask)
        ctx.cfg = (heads, float(p_drop), int(seed))
        if qkv_lo is not None:      # two-term forward: (qkv, qkv_lo) -> (out, out_lo); the backward reads the hi planes
            assert qkv.dtype == torch.bfloat16 and qkv_lo.is_contiguous()
            pair = Pair.empty((B, T, HD), qkv.device)
            bits = _attn_fwd(q, k, v, pair.hi, lse, B, T, heads, HD // heads, HD3, p_drop, seed, mask,
                             lo_off=(qkv_lo.data_ptr() - qkv.data_ptr()) // 2, out_lo_off=pair.hi.numel())
            ctx.save_for_backward(qkv, pair.hi, lse, bits, mask)
            ctx.mark_non_differentiable(pair.lo)
            return pair.hi, pair.lo
        out = torch.empty((B, T, HD), dtype=qkv.dtype, device=qkv.device)
        bits = _attn_fwd(q, k, v, out, lse, B, T, heads, HD // heads, HD3, p_drop, seed, mask)
        ctx.save_for_backward(qkv, out, lse, bits, mask)
        return out

    @staticmethod
    def backward(ctx, dout, *_):
        qkv, out, lse, bits, mask = ctx.saved_tensors
        heads, p_drop, seed = ctx.cfg
        B, T, HD3 = qkv.shape
        HD = HD3 // 3
        dqkv = torch.empty_like(qkv)
        _attn_bwd(qkv[..., :HD], qkv[..., HD:2 * HD], qkv[..., 2 * HD:], out, dout.to(qkv.dtype).contiguous(), lse, bits,
                  dqkv[..., :HD], dqkv[..., HD:2 * HD], dqkv[..., 2 * HD:], B, T, heads, HD // heads, HD3,
                  p_drop, seed, mask)
        return dqkv, None, None, None, None, None


EPI_DROPOUT_BF16, EPI_MASK_BF16 = 1, 2


def _gemm_epilogue(a, b, out, bias, mode, relu=0, p_drop=0.0, seed=0, aux=None, scale=1.0):
    M, K = a.shape
    N = b.shape[0]
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().pika_gemm_bf16_epilogue(
            a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K,
            None if bias is None else bias.data_ptr(), mode, relu, float(p_drop), int(seed),
            None if aux is None else aux.data_ptr(), 0 if aux is None else aux.stride(0), float(scale),
            _stream()), "pika_gemm_bf16_epilogue(M=%d,N=%d,K=%d)" % (M, N, K))
    return out


def dropout_keep_mask(rows, cols, p_drop, seed, device):
    """(rows, cols) bool keep-mask of the PIKA_EPI_DROPOUT_BF16 epilogue -- for tests."""
    m = torch.empty((rows, cols), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().pika_dropout_keep_mask(m.data_ptr(), rows, cols, float(p_drop), int(seed), _stream()),
                   "pika_dropout_keep_mask")
    return m.bool()


def _gemm_dropout_residual(a, b, out, bias, p_drop, seed, residual):
    M, K = a.shape
    N = b.shape[0]
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().pika_gemm_bf16_dropout_residual(
            a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K,
            None if bias is None else bias.data_ptr(), float(p_drop), int(seed), residual.data_ptr(),
            residual.stride(0), _stream()), "pika_gemm_bf16_dropout_residual(M=%d,N=%d,K=%d)" % (M, N, K))
    return out


def _mask_cast(x2d, p_drop, seed):
    """bf16 copy of a gradient with the dropout backward (same hash as the forward epilogue) applied."""
    out = torch.empty(x2d.shape, dtype=torch.bfloat16, device=x2d.device)
    with torch.cuda.device(x2d.device):
        _lib.check(_lib.lib().pika_dropout_mask_cast_bf16(
            x2d.data_ptr(), x2d.stride(0), x2d.shape[0], x2d.shape[1], float(p_drop), int(seed), out.data_ptr(),
            out.stride(0), _stream()), "pika_dropout_mask_cast_bf16")
    return out


def linear_dropout_residual_ok(x, weight, residual):
    N, K = weight.shape
    return (G.bf16_backward() and _fused() and x.is_cuda and x.dtype == torch.bfloat16 and residual.dtype == torch.float32
            and K % 64 == 0 and N % 64 == 0 and residual.shape[-1] == N and residual.is_contiguous())


class LinearDropoutResidualFn(torch.autograd.Function):
    """dropout(x @ W^T + b) + residual in ONE GEMM (x bf16, output fp32): the attention output projection with
    the layer's residual dropout and add (transformer.py:98-99).  Backward: the dropout mask is re-applied
    while the incoming gradient is rounded to bf16 for the dX / dW products; the residual gets it as is."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, p_drop, seed, x_lo=None):
        N, K = weight.shape
        x2 = x.reshape(-1, K).contiguous()
        out = torch.empty(residual.shape, dtype=torch.float32, device=x.device)
        if x_lo is not None:        # two-term forward product
            with torch.cuda.device(x.device):
                hi, lo, Cp = _planes(x, x_lo, K)
                G.gemm_ex(G.pair_operand(hi, lo, hi.shape[0], Cp), G.split_weight(weight), hi.shape[0], N, 3 * Cp,
                          G.EPI_DROPOUT_RESIDUAL, out.view(-1, N), bias=bias, p_drop=p_drop, seed=seed,
                          residual=residual.view(-1, N))
        else:
            _gemm_dropout_residual(x2, weight.detach().to(torch.bfloat16), out.view(-1, N), bias, p_drop, seed,
                                   residual.view(-1, N))
        ctx.cfg = (float(p_drop), int(seed), x.shape)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x2, weight)
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, weight = ctx.saved_tensors
        p_drop, seed, xshape = ctx.cfg
        N, K = weight.shape
        d2 = dout.reshape(-1, N).contiguous()
        M = d2.shape[0]
        dx = dw = db = None
        with torch.cuda.device(dout.device):
            dyb = _mask_cast(d2, p_drop, seed)
            if ctx.needs_input_grad[0]:
                dx = torch.empty(xshape, dtype=torch.bfloat16, device=dout.device)
                _gemm_epilogue(dyb, _weight_t(weight), dx.view(-1, K), None, EPI_DROPOUT_BF16)
            if ctx.needs_input_grad[1]:
                dw = _grad_weight(dyb, G.matrix(x2)[0], 8, M, K, N)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = colsum_any(dyb)
        return dx, dw, db, (dout if ctx.needs_input_grad[3] else None), None, None, None


def feed_forward_ok(x, w1, w2):
    d, f = w1.shape[1], w1.shape[0]
    return (G.bf16_backward() and _fused() and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and d % 64 == 0 and f % 64 == 0
            and w2.shape[0] % 64 == 0 and x.numel() // d >= 256)


class FeedForwardFn(torch.autograd.Function):
    """w_2(dropout(relu(w_1(x))))  (position_ffn.py:27-39 without the LayerNorm and the residual).

    The (rows, d_ff) hidden exists only as ONE bf16 matrix: the first GEMM's epilogue applies bias, ReLU and
    dropout (counter-based mask) and rounds to bf16, which is what the second GEMM would read anyway; in the
    backward `hidden > 0` is both the ReLU and the dropout mask, applied by the epilogue of dh = dy W2."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, p_drop, seed, residual=None, p2=0.0, seed2=0, x_lo=None):
        d = x.shape[-1]
        ctx.x_bf16 = x.dtype == torch.bfloat16
        if _mixed() and (x_lo is not None or x.dtype == torch.float32):
            # both products in two terms; the hidden exists as two bf16 planes written by the first product's epilogue
            F, N2 = w1.shape[0], w2.shape[0]
            with torch.cuda.device(x.device):
                xh, xl, Cp = _planes(x, x_lo, d)
                M = xh.shape[0]
                hp = Pair.empty((M, F), x.device)
                G.gemm_ex(G.pair_operand(xh, xl, M, Cp), G.split_weight(w1), M, F, 3 * Cp, G.EPI_DROPOUT_BF16, hp.hi,
                          out_lo=hp.lo, bias=b1, relu=True, p_drop=p_drop, seed=seed)
                y = torch.empty(x.shape[:-1] + (N2,), dtype=torch.float32, device=x.device)
                a2 = G.pair_operand(hp.hi, hp.lo, M, F)
                if residual is not None:
                    G.gemm_ex(a2, G.split_weight(w2), M, N2, 3 * F, G.EPI_DROPOUT_RESIDUAL, y.view(-1, N2), bias=b2,
                              p_drop=p2, seed=seed2, residual=residual.contiguous().view(-1, N2))
                else:
                    G.gemm_ex(a2, G.split_weight(w2), M, N2, 3 * F, G.EPI_F32, y.view(-1, N2), bias=b2)
            ctx.res = (residual is not None, float(p2), int(seed2))
            ctx.cfg = (float(p_drop), x.shape)
            ctx.save_for_backward(xh, hp.hi, w1, w2)
            return y
        xb = x.reshape(-1, d).to(torch.bfloat16).contiguous()
        M, F, N2 = xb.shape[0], w1.shape[0], w2.shape[0]
        h = torch.empty((M, F), dtype=torch.bfloat16, device=x.device)
        _gemm_epilogue(xb, w1.detach().to(torch.bfloat16), h, b1, EPI_DROPOUT_BF16, relu=1, p_drop=p_drop, seed=seed)
        y = torch.empty(x.shape[:-1] + (N2,), dtype=torch.float32, device=x.device)
        if residual is not None:   # dropout_2(w_2(.)) + x of position_ffn.py:38-39 in the second GEMM's epilogue
            _gemm_dropout_residual(h, w2.detach().to(torch.bfloat16), y.view(-1, N2), b2, p2, seed2,
                                   residual.contiguous().view(-1, N2))
        else:
            G.gemm_bf16_nt(h, w2.detach().to(torch.bfloat16), bias=b2, out=y.view(-1, N2))
        ctx.res = (residual is not None, float(p2), int(seed2))
        ctx.cfg = (float(p_drop), x.shape)
        ctx.save_for_backward(xb, h, w1, w2)
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, h, w1, w2 = ctx.saved_tensors
        p_drop, xshape = ctx.cfg
        M, F, N2, d = xb.shape[0], w1.shape[0], w2.shape[0], xb.shape[1]
        dy2 = dy.reshape(-1, N2).contiguous()
        has_res, p2, seed2 = ctx.res
        dyb = _mask_cast(dy2, p2, seed2) if has_res else dy2.to(torch.bfloat16)
        thr = round(p_drop * 65536)
        dh = torch.empty((M, F), dtype=torch.bfloat16, device=dy.device)
        _gemm_epilogue(dyb, weight_taps_transposed(w2), dh, None, EPI_MASK_BF16, aux=h,
                       scale=65536.0 / (65536 - thr))
        dx = dw1 = db1 = dw2 = db2 = None
        with torch.cuda.device(dy.device):
            if ctx.needs_input_grad[3]:
                dw2 = _grad_weight(dyb, G.matrix(h)[0], 8, M, F, N2)
            if ctx.needs_input_grad[4]:
                db2 = colsum_any(dyb) if has_res else colsum(dy2)
            if ctx.needs_input_grad[1]:
                dw1 = _grad_weight(dh, G.matrix(xb)[0], 8, M, d, F)
            if ctx.needs_input_grad[2]:
                db1 = torch.empty(F, dtype=torch.float32, device=dy.device)
                _colsum_into(db1, dh.data_ptr(), F, M, F, dy.device, True)
            if ctx.needs_input_grad[0]:
                w1t = weight_taps_transposed(w1)
                if ctx.x_bf16:
                    dx = torch.empty(xshape, dtype=torch.bfloat16, device=dy.device)
                    _gemm_epilogue(dh, w1t, dx.view(-1, d), None, EPI_DROPOUT_BF16)
                else:
                    dx = G.gemm_bf16_nt(dh, w1t).view(xshape)
        dres = dy if (has_res and ctx.needs_input_grad[7]) else None
        return dx, dw1, db1, dw2, db2, None, None, dres, None, None, None


def _dt(t):
    return G.PIKA_F32 if t.dtype == torch.float32 else G.PIKA_BF16


class BnValid(ctypes.Structure):
    """include/pika_norm.h: pika_bn_valid_t."""
    _fields_ = [("t_valid", ctypes.c_void_p), ("rows_per_batch", ctypes.c_int), ("sub", ctypes.c_int),
                ("div", ctypes.c_int)]


_VALID_ROWS = None      # (t_valid (1,) int32 device tensor, rows_per_batch, sub, div) while an encoder layer runs on a batch
#                         whose time axis is padded beyond its data (pika_amd/train_graph.py); None: every row counts


class valid_rows(object):
    """Context: the BatchNorm launches issued inside count only the first (*t_valid - sub) // div rows of every block of
    rows_per_batch rows (statistics, row count, backward), write zeros to the rest and send no gradient there."""

    def __init__(self, t_valid, rows_per_batch, sub, div):
        self.v = None if t_valid is None else (t_valid, int(rows_per_batch), int(sub), int(div))

    def __enter__(self):
        global _VALID_ROWS
        self.old, _VALID_ROWS = _VALID_ROWS, self.v

    def __exit__(self, *exc):
        global _VALID_ROWS
        _VALID_ROWS = self.old


def _valid_arg(valid, rows):
    if valid is None:
        return None
    t, rpb, sub, div = valid
    assert rows % rpb == 0 and t.dtype == torch.int32 and t.is_cuda, (rows, rpb)
    return ctypes.byref(BnValid(t.data_ptr(), rpb, sub, div))


def _bn_forward(x, weight, bias, running_mean, running_var, eps, momentum, out_bf16, out_pair=False, valid=None):
    """Training-mode statistics + apply; returns (y, mean, rstd); with out_pair y is a Pair (two bf16 planes)."""
    M, C = x.shape
    lib = _lib.lib()
    stats = torch.empty(2 * C, dtype=torch.float64, device=x.device)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    pair = Pair.empty(x.shape, x.device) if out_pair else None
    y = pair.hi if out_pair else torch.empty(x.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    _lib.check(lib.pika_bn_stats(x.data_ptr(), M, C, stats.data_ptr(), _valid_arg(valid, M), _stream()), "pika_bn_stats")
    _lib.check(lib.pika_bn_apply(
        x.data_ptr(), M, C, stats.data_ptr(), weight.data_ptr(), bias.data_ptr(), float(eps),
        float(momentum), None if running_mean is None else running_mean.data_ptr(),
        None if running_var is None else running_var.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
        y.data_ptr(), _dt(y), pair.lo.data_ptr() if out_pair else None, _valid_arg(valid, M), _stream()), "pika_bn_apply")
    return (pair if out_pair else y), mean, rstd


def _bn_backward(dy, x, weight, mean, rstd, relu_mask, dx_bf16, valid=None):
    """Returns (dx, dgamma, dbeta); dy f32 | bf16, dx f32 | bf16."""
    M, C = x.shape
    if dy.dtype not in (torch.float32, torch.bfloat16):
        dy = dy.float()
    dy = dy.contiguous()
    sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
    dx = torch.empty(x.shape, dtype=torch.bfloat16 if dx_bf16 else torch.float32, device=x.device)
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty_like(dg)
    _lib.check(_lib.lib().pika_bn_backward(dy.data_ptr(), _dt(dy), x.data_ptr(), M, C, weight.data_ptr(),
                                           mean.data_ptr(), rstd.data_ptr(), sums.data_ptr(),
                                           dx.data_ptr(), _dt(dx), dg.data_ptr(), db.data_ptr(),
                                           int(relu_mask), _valid_arg(valid, M), _stream()), "pika_bn_backward")
    return dx, dg, db


class BatchNormFn(torch.autograd.Function):
    """Training-mode BatchNorm1d over the rows of a (M,C) matrix (include/pika_norm.h); running
    statistics updated in place exactly as nn.BatchNorm1d does (momentum, unbiased variance).
    out_bf16: the result only feeds an MFMA product (the next time-delay layer)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, relu_input=False, out_bf16=False,
                out_pair=False):
        x = x.contiguous()
        ctx.relu_input = bool(relu_input)
        ctx.valid = _VALID_ROWS
        ctx.set_materialize_grads(False)
        with torch.cuda.device(x.device):
            y, mean, rstd = _bn_forward(x, weight, bias, running_mean, running_var, eps, momentum, out_bf16, out_pair,
                                        valid=ctx.valid)
        ctx.save_for_backward(x, weight, mean, rstd)
        if out_pair:
            ctx.mark_non_differentiable(y.lo)
            return y.hi, y.lo
        return y

    @staticmethod
    def backward(ctx, dy, *_):
        x, weight, mean, rstd = ctx.saved_tensors
        with torch.cuda.device(x.device):
            dx, dg, db = _bn_backward(dy, x, weight, mean, rstd, ctx.relu_input, False, valid=ctx.valid)
        return dx, dg, db, None, None, None, None, None, None, None


class TdnnBnFn(torch.autograd.Function):
    """bn(relu(time_delay(x)))  (rnnt_tdnn_transformer.py:76-82) as one autograd node: GEMM with bias + ReLU
    epilogue -> fp32 y -> BatchNorm (batch statistics).  Inside the node the gradient of y never exists in
    fp32: the BatchNorm backward applies the ReLU mask and writes it as the bf16 matrix the dX / dW products
    read; with out_bf16 the output (and its incoming gradient) is bf16 too."""

    @staticmethod
    def forward(ctx, x, w2d, bias, taps, dil, stride, pad, bn_w, bn_b, running_mean, running_var, eps, momentum,
                out_bf16, x_lo=None, out_pair=False):
        ctx.x_bf16 = x.dtype == torch.bfloat16
        ctx.valid = _VALID_ROWS
        ctx.set_materialize_grads(False)
        Bn, T, C = x.shape
        N = w2d.shape[0]
        if _mixed() and (x_lo is not None or x.dtype == torch.float32):
            # two-term forward product over a time-delay view of the two planes; y, the BatchNorm input, stays fp32
            with torch.cuda.device(x.device):
                hi, lo, Cp = _planes(x.contiguous(), x_lo, C)
                t_out = G.time_delay(x, taps, dil, stride, pad)[3]
                M = Bn * t_out
                a_op = G.pair_operand(hi, lo, M, Cp, taps, dil, stride, pad, rows_per_batch=t_out, t_in=T,
                                      batch_stride=T * Cp)
                y = torch.empty((M, N), dtype=torch.float32, device=x.device)
                G.gemm_ex(a_op, G.split_weight(w2d, taps), M, N, taps * 3 * Cp, G.EPI_F32, y, bias=bias, relu=True)
                out, mean, rstd = _bn_forward(y, bn_w, bn_b, running_mean, running_var, eps, momentum, out_bf16, out_pair,
                                              valid=ctx.valid)
            xb = hi.view(Bn, T, Cp)
        else:
            xb = x.contiguous() if ctx.x_bf16 else x.contiguous().to(torch.bfloat16)
            with torch.cuda.device(x.device):
                a_op, M, K, t_out = G.time_delay(xb, taps, dil, stride, pad)
                y = torch.empty((M, N), dtype=torch.float32, device=x.device)
                G.launch(a_op, G.matrix(_weight_for(xb, w2d))[0], y, N, M, N, K, bias=bias, relu=True)
                out, mean, rstd = _bn_forward(y, bn_w, bn_b, running_mean, running_var, eps, momentum, out_bf16,
                                              valid=ctx.valid)
        ctx.cfg = (taps, dil, stride, pad, t_out)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(xb, w2d, y, bn_w, mean, rstd)
        if isinstance(out, Pair):
            o_hi, o_lo = out.hi.view(Bn, t_out, N), out.lo.view(Bn, t_out, N)
            ctx.mark_non_differentiable(o_lo)
            return o_hi, o_lo
        return out.view(Bn, t_out, N)

    @staticmethod
    def backward(ctx, dout, *_):
        xb, w2d, y, bn_w, mean, rstd = ctx.saved_tensors
        taps, dil, stride, pad, t_out = ctx.cfg
        Bn, T, C = xb.shape
        N, K = w2d.shape
        M = y.shape[0]
        dx = dw = db = None
        with torch.cuda.device(dout.device):
            dyb, dg, dbeta = _bn_backward(dout.reshape(M, N), y, bn_w, mean, rstd, True, True, valid=ctx.valid)
            if ctx.needs_input_grad[0]:
                # bf16 dx straight from the epilogue when the direct-to-LDS kernel takes the product (its own gate)
                direct = (ctx.x_bf16 and stride == 1 and N % 64 == 0 and C >= 192 and C % 4 == 0
                          and ((Bn * T + 255) // 256) * ((C + 255) // 256) >= 160)
                dx = torch.empty(xb.shape, dtype=torch.bfloat16 if direct else torch.float32, device=dout.device)
                if stride == 1 and N % 64 == 0:
                    wrev = weight_taps_transposed(w2d, taps)
                    a_op = G.Operand(dyb.data_ptr(), G.PIKA_BF16, T, t_out, t_out * N, N, N, 1, dil,
                                     (taps - 1) * dil - pad, 0, 0)
                    G.launch(a_op, G.matrix(wrev)[0], dx.view(-1, C), C, Bn * T, C, taps * N)
                else:
                    dcol = _grad_input(dyb, w2d)  # (M, taps*C)
                    _lib.check(_lib.lib().pika_col2im(dcol.data_ptr(), dx.data_ptr(), Bn, t_out, T, C,
                                                      taps, stride, dil, pad, _stream()), "pika_col2im")
                if ctx.x_bf16 and dx.dtype != torch.bfloat16:
                    dx = dx.to(torch.bfloat16)
            if ctx.needs_input_grad[1]:
                a_op = G.time_delay(xb, taps, dil, stride, pad)[0]
                dw = _grad_weight(dyb, a_op, _g(xb), M, K, N)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = colsum_any(dyb)
        return dx, dw, db, None, None, None, None, dg, dbeta, None, None, None, None, None, None, None
