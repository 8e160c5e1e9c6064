"""Host wrapper of the MFMA GEMM (include/pika_gemm.h): builds operand descriptors from torch
tensors (device memory + strides only) and launches on torch's current stream."""
import ctypes
import os

import torch

from . import _lib

PIKA_F32, PIKA_BF16 = 0, 1
RELU, ACCUMULATE, FP32SPLIT = 1, 2, 4

# "bf16":   one MFMA per product (config-2 arithmetic).
# "bf16x3": every fp32 operand as two bf16 terms, hi.hi + lo.hi + hi.lo as ONE bf16 product over a three times
#           longer reduction (include/pika_ops.h: pika_split_bf16_terms) -- ~1e-5 relative per product, on the
#           direct-to-LDS kernels; tensors stay fp32 between products exactly as in the "fp32" mode.
# "fp32":   exact three-term split, the six products above 2^-24 of the leading one (~1e-7; parity runs, decode):
#           products large enough for the direct-to-LDS kernels take the same K-concatenation with six segments,
#           everything else the register-staged kernel that issues the 6 MFMAs per loaded tile.
# "mixed":  the train-step default that carries the parity statement of north_star (encoder activations and loss within
#           1e-3 of the reference's fp32): FORWARD products of the encoder, the prediction network and the joint's
#           projections in two bf16 terms per operand (hi.hi + lo.hi + hi.lo on the direct-to-LDS kernel through a
#           segment map, pika_operand_t.seg: activations travel between products as two bf16 planes "hi" / "lo" written by
#           the producing epilogue, LayerNorm, BatchNorm or attention kernel), the fused attention forward in the same
#           arithmetic; the joint's lattice products and EVERY backward product on one bf16 term, reading the hi planes.
#           Exact forward ReLU / dropout masks, bf16 gradients.
# Overridable per call.  The default is "mixed": what a training script started through `python -m pika_amd.launch` gets
# is the arithmetic whose encoder activations and loss are within 1e-3 of the reference's (tests/test_model_full.py);
# PIKA_GEMM_PRECISION=bf16 buys 15 % of step time at 3e-2.
PRECISION = os.environ.get("PIKA_GEMM_PRECISION", "mixed")
PRECISIONS = ("bf16", "bf16x3", "fp32", "mixed", "fp16x2")
# "fp16x2" (inference products: the decoder's encoder pass, the rescorer): every fp32 operand as two FP16 terms (22 mantissa
#           bits), hi.hi + lo.hi + hi.lo as ONE fp16 product over a three times longer reduction (pika_split_bf16_terms
#           n_terms = 4: power-of-two factors keep the small terms normal and cancel in the products) -- an fp32 product to
#           ~2^-22 at half the cost of the exact six-segment path; products the direct-to-LDS kernel does not take run exact.
# "bf16x3": the joint's lattice products (fc2 over the (B,T,U) lattice and its two gradient products: half of a training
# step's FLOPs, on a hidden the gate kernel writes once) stay in the config-2 bf16 arithmetic by default -- the
# encoder, the prediction network and the joint's projections are what the parity statement (encoder activations,
# loss) rests on, and the loss moves by ~1e-5 (profiles/r2_precision_table.md).  X3_JOINT_BF16 = False: two terms there too.
X3_JOINT_BF16 = True


def joint_in_bf16():
    """The joint's lattice products run on bf16 operands: mode "bf16", or "bf16x3" with the default above."""
    return PRECISION in ("bf16", "mixed") or (PRECISION == "bf16x3" and X3_JOINT_BF16)


def bf16_backward():
    """Modes whose backward products run on one bf16 term (and whose activations between MFMA products are bf16 planes)."""
    return PRECISION in ("bf16", "mixed")
FP16X2_STATS = {"fast": 0, "exact": 0}     # "fp16x2" products on the three-segment fp16 path / handed to the exact path
BF16X3_STATS = {"fast": 0, "exact": 0}     # "bf16x3" products taken by the split path / handed to the exact path
FP32_STATS = {"concat": 0, "staged": 0}    # "fp32" products on the six-segment path / on the register-staged kernel


class Operand(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("dtype", ctypes.c_int), ("rows_per_batch", ctypes.c_int),
                ("t_in", ctypes.c_int), ("batch_stride", ctypes.c_longlong), ("ld", ctypes.c_longlong),
                ("C", ctypes.c_int), ("stride", ctypes.c_int), ("dil", ctypes.c_int),
                ("pad", ctypes.c_int), ("z_outer", ctypes.c_longlong), ("z_inner", ctypes.c_longlong),
                ("trans", ctypes.c_int), ("seg", ctypes.c_int), ("lo_off", ctypes.c_longlong)]


def _dtype(t):
    if t.dtype == torch.float32:
        return PIKA_F32
    if t.dtype == torch.bfloat16:
        return PIKA_BF16
    raise TypeError("pika_amd.gemm: operands must be float32 or bfloat16, got %s" % t.dtype)


def matrix(t, z_outer=0, z_inner=0):
    """Plain row-major matrix view (rows, K) with unit inner stride."""
    assert t.dim() == 2 and t.stride(1) == 1, "need (rows,K) with contiguous K"
    rows, K = t.shape
    return Operand(t.data_ptr(), _dtype(t), max(rows, 1), max(rows, 1), 0, t.stride(0), K, 1, 0, 0,
                   z_outer, z_inner), rows, K


def matrix_t(t):
    """The (rows,K) matrix `t` used TRANSPOSED: output index = its columns, reduction = its rows.
    Returns (operand, n_out, n_red)."""
    op, rows, K = matrix(t)
    op.trans = 1
    return op, K, rows


def time_delay(x, taps, dil=1, stride=1, pad=0):
    """Virtual (B*T_out, taps*C) matrix over x (B,T,C): row (b,t), col (tap,c) reads
    x[b, t*stride + tap*dil - pad, c]."""
    assert x.dim() == 3 and x.stride(2) == 1
    Bn, T, C = x.shape
    t_out = (T + pad - dil * (taps - 1) - 1) // stride + 1 if pad == 0 else T
    op = Operand(x.data_ptr(), _dtype(x), t_out, T, x.stride(0), x.stride(1), C, stride, dil, pad, 0, 0)
    return op, Bn * t_out, taps * C, t_out


def _flags(relu, accumulate, precision):
    p = precision or PRECISION
    if p not in PRECISIONS:
        raise ValueError("unknown GEMM precision %r" % (p,))
    if p == "mixed":
        p = "bf16"      # "mixed" forward products come through gemm_ex; what reaches pika_gemm_nt is the backward
    # "bf16x3" reaches here only for products its operand split does not take (launch): those run exactly
    return (RELU if relu else 0) | (ACCUMULATE if accumulate else 0) | (FP32SPLIT if p != "bf16" else 0)


OUT_BF16 = 8
F16_OPERANDS = 16
TERM_PRODUCT = 32
# "fp32" products of direct-to-LDS size as ONE bf16 product over six term segments (FP32_CONCAT = False: always the
# register-staged exact kernel)
FP32_CONCAT = True
FP32_CONCAT_MAX_BYTES = 4 << 30     # per operand copy: 6 segments x 2 bytes x rows x reduction


_WORKSPACE = {}
WORKSPACE_BYTES = 160 << 20


def _workspace(device):
    """Per-device scratch for split-K partial tiles (stream-ordered use on the current stream)."""
    ws = _WORKSPACE.get(device.index)
    if ws is None:
        ws = _WORKSPACE[device.index] = torch.empty(WORKSPACE_BYTES, dtype=torch.uint8, device=device)
    return ws


def _pad64(n):
    return (n + 63) & ~63


class SplitJob(ctypes.Structure):
    """pika_split_job_t (include/pika_ops.h)"""
    _fields_ = [("x", ctypes.c_void_p), ("n_batch", ctypes.c_int), ("t_in", ctypes.c_int), ("C", ctypes.c_int),
                ("batch_stride", ctypes.c_longlong), ("ld", ctypes.c_longlong), ("role", ctypes.c_int),
                ("n_terms", ctypes.c_int), ("layout", ctypes.c_int), ("Cp", ctypes.c_int), ("dst", ctypes.c_void_p)]


def _split(op, n_batch, t_in, C, batch_stride, ld, role, layout, Cp, device, n_terms=2, pending=None):
    """bf16 term-segment copy of an f32 source (pika_split_bf16_terms: 3 segments for two terms, 6 for three);
    returns the tensor (flat).  pending (a list): the launch is left to `_split_flush`, which sends the two operands of a
    product as ONE launch."""
    rows = n_batch * t_in
    nseg = 6 if n_terms == 3 else 3
    dst = torch.empty(nseg * rows * Cp, dtype=torch.bfloat16, device=device)
    if pending is not None:
        pending.append(SplitJob(op.ptr, n_batch, t_in, C, batch_stride, ld, role, n_terms, layout, Cp, dst.data_ptr()))
        return dst
    rc = _lib.lib().pika_split_bf16_terms(op.ptr, n_batch, t_in, C, batch_stride, ld, role, n_terms, layout, Cp,
                                          dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "pika_split_bf16_terms")
    return dst


def _split_flush(pending):
    if not pending:
        return
    st = torch.cuda.current_stream().cuda_stream
    lib = _lib.lib()
    if len(pending) == 2:
        _lib.check(lib.pika_split_bf16_terms2(ctypes.byref(pending[0]), ctypes.byref(pending[1]), st), "pika_split_bf16_terms2")
    else:
        for j in pending:
            _lib.check(lib.pika_split_bf16_terms(j.x, j.n_batch, j.t_in, j.C, j.batch_stride, j.ld, j.role, j.n_terms, j.layout,
                                                 j.Cp, j.dst, st), "pika_split_bf16_terms")
    del pending[:]


def _plain(op, K):
    return op.C >= K and op.pad == 0 and op.stride == 1


MIN_TILES = 160
TERM_MIN_TILES = 24      # ... of a K-concatenated term product (include/pika_gemm.h: PIKA_GEMM_TERM_PRODUCT)


def set_min_tiles(n):
    """The direct-to-LDS kernels' fill-the-chip gate (include/pika_gemm.h: pika_gemm_set_min_tiles), here and in the
    library.  Returns the previous value.  Tests lower it so that a B = 4 golden runs the K-concatenated term products
    the benchmarked batch sizes run."""
    global MIN_TILES
    old, MIN_TILES = MIN_TILES, int(n)
    _lib.lib().pika_gemm_set_min_tiles(int(n))
    return old


def _direct_to_lds_size(trans, M, N, K, terms=False):
    """The size gates of the direct-to-LDS kernels (gemm_glds.hip: pika_internal_gemm_pp / launch_pp_tn); terms: of a
    K-concatenated term product (launched with TERM_PRODUCT)."""
    if trans:
        return M >= 192 and N >= 192 and K >= 512
    return M >= 256 and N >= 192 and ((M + 255) // 256) * ((N + 255) // 256) >= (min(MIN_TILES, TERM_MIN_TILES) if terms
                                                                                 else MIN_TILES)


def _bf16x3_operands(a_op, b_op, M, N, K, device, n_terms=2):
    """The two operands of a K-concatenated product as bf16 operands over a reduction of S = 3 (two terms) or 6 (three
    terms) segments, or None when the split does not apply (bf16 / batched / mixed-orientation operands, extents the
    16-byte bf16 loads cannot take): the caller then runs the exact path.  Returns (a3, b3, K3, keep-alive tensors)."""
    S = 6 if n_terms == 3 else 3
    if a_op.dtype != PIKA_F32 or b_op.dtype != PIKA_F32 or bool(a_op.trans) != bool(b_op.trans):
        return None
    if n_terms == 4 and a_op.trans:
        return None             # two fp16 terms: forward (reduction-contiguous) products only
    if a_op.z_outer or a_op.z_inner or b_op.z_outer or b_op.z_inner:
        return None
    if not a_op.trans:
        # reduction index contiguous: segments side by side inside every (tap) block of the reduction
        if a_op.C < K and b_op.C < K:
            return None
        seg = min(a_op.C, b_op.C, K)
        if seg & 7 or K % seg:
            return None
        taps, Cp = K // seg, _pad64(seg)
        ops, keep, pending = [], [], []
        for op, extent, role in ((a_op, M, 0), (b_op, N, 1)):
            if op.C < K or taps == 1:       # the time-delay view itself, or a plain matrix with one segment per row
                nb = (extent + op.rows_per_batch - 1) // op.rows_per_batch
                t = _split(op, nb, op.t_in, seg, op.batch_stride, op.ld, role, 0, Cp, device, n_terms, pending)
                # one block: batch stride 0 like every plain matrix (the direct-to-LDS kernel bounds its 32-bit row
                # offsets by rows_per_tile * pitch + batch stride)
                new = Operand(t.data_ptr(), PIKA_BF16, op.rows_per_batch, op.t_in, op.t_in * S * Cp if nb > 1 else 0,
                              S * Cp, S * Cp, op.stride, op.dil, op.pad, 0, 0)
                if op.C >= K:
                    new.C = S * Cp * taps
            else:                           # plain (rows, taps*seg) matrix against a time-delay view: per-tap segments
                if op.ld != K or not _plain(op, K):
                    return None                     # (a first operand's pending split is dropped with it)
                t = _split(op, 1, extent * taps, seg, 0, seg, role, 0, Cp, device, n_terms, pending)
                rows = max(extent, 1)
                new = Operand(t.data_ptr(), PIKA_BF16, rows, rows, 0, taps * S * Cp, taps * S * Cp, 1, 0, 0, 0, 0)
            ops.append(new)
            keep.append(t)
        _split_flush(pending)
        return ops[0], ops[1], taps * S * Cp, keep
    # `trans` operands (dW = dY^T X): the reduction runs over the rows -> the three segments are stacked row blocks
    ops, keep, pending = [], [], []
    for op, extent, role in ((a_op, M, 0), (b_op, N, 1)):
        nb = (K + op.rows_per_batch - 1) // op.rows_per_batch
        if op.C & 7 or extent & 7 or op.pad or nb * op.rows_per_batch != K:
            return None
        t = _split(op, nb, op.t_in, op.C, op.batch_stride, op.ld, role, 1, op.C, device, n_terms, pending)
        if nb == 1 and op.t_in == op.rows_per_batch:    # one block: the stack is one plain (S*R, C) matrix
            new = Operand(t.data_ptr(), PIKA_BF16, S * K, S * K, 0, op.C, op.C, op.stride, op.dil, 0, 0, 0)
        else:
            new = Operand(t.data_ptr(), PIKA_BF16, op.rows_per_batch, op.t_in, op.t_in * op.C, op.C, op.C,
                          op.stride, op.dil, 0, 0, 0)
        new.trans = 1
        ops.append(new)
        keep.append(t)
    _split_flush(pending)
    return ops[0], ops[1], S * K, keep


def launch(a_op, b_op, out, ldc, M, N, K, bias=None, relu=False, accumulate=False, precision=None,
           batch=1, z_div=1, c_z_outer=0, c_z_inner=0):
    if not out.is_cuda:
        raise RuntimeError("pika_amd.gemm: tensors must live on a HIP device (no CPU path)")
    keep = None
    p = precision or PRECISION
    if p == "mixed":
        p = precision = "bf16"
    if a_op.dtype == PIKA_BF16 and b_op.dtype == PIKA_BF16:
        p = precision = "bf16"              # a product of bf16 operands has nothing to split, whatever the mode
    elif p != "bf16" and (a_op.dtype == PIKA_BF16 or b_op.dtype == PIKA_BF16) and (a_op.trans or b_op.trans):
        # one bf16 operand against a reduction-major fp32 one (a weight gradient whose activation was stored in bf16):
        # the exact kernel does not take that pairing; the stored operand already carries the bf16 rounding
        p = precision = "bf16"
    f16 = term = False
    if p == "fp16x2":
        ok = (batch == 1 and not c_z_outer and not c_z_inner and out.dtype == torch.float32 and not accumulate
              and not a_op.trans and not b_op.trans and _direct_to_lds_size(False, M, N, K, terms=True))
        with torch.cuda.device(out.device):
            sp = _bf16x3_operands(a_op, b_op, M, N, K, out.device, 4) if ok else None
        if sp is not None:
            a_op, b_op, K, keep = sp
            p = precision = "bf16"
            f16 = term = True
            FP16X2_STATS["fast"] += 1
        else:
            p = precision = "fp32"      # small / transposed / batched products: exact
            FP16X2_STATS["exact"] += 1
    if p == "bf16x3" or (p == "fp32" and not accumulate and FP32_CONCAT):
        n_terms = 2 if p == "bf16x3" else 3
        splittable = batch == 1 and not c_z_outer and not c_z_inner and out.dtype == torch.float32
        if n_terms == 3:    # only where the direct-to-LDS kernels will take the product (6x their time beats 6 MFMAs
            #                 per tile on the register-staged kernel) and the six-segment copies stay moderate (the
            #                 joint's lattice-sized operands would need 2 x 24 GB of temporaries per product)
            splittable = (splittable and bool(a_op.trans) == bool(b_op.trans) and _direct_to_lds_size(a_op.trans, M, N, K, terms=True)
                          and 12 * max(M, N) * K <= FP32_CONCAT_MAX_BYTES)
        with torch.cuda.device(out.device):
            sp = _bf16x3_operands(a_op, b_op, M, N, K, out.device, n_terms) if splittable else None
        stats, hit, miss = (BF16X3_STATS, "fast", "exact") if n_terms == 2 else (FP32_STATS, "concat", "staged")
        if sp is not None:
            a_op, b_op, K, keep = sp        # `keep` holds the split copies until the launch below is enqueued
            precision = "bf16"
            term = True
            stats[hit] += 1
        else:
            stats[miss] += 1
    with torch.cuda.device(out.device):
        ws = _workspace(out.device) if (a_op.trans and b_op.trans) else None
        rc = _lib.lib().pika_gemm_nt_ws(ctypes.byref(a_op), ctypes.byref(b_op), out.data_ptr(), ldc,
                                        c_z_outer, c_z_inner, M, N, K, batch, z_div,
                                        None if bias is None else bias.data_ptr(),
                                        _flags(relu, accumulate, precision) | (OUT_BF16 if out.dtype == torch.bfloat16 else 0)
                                        | (F16_OPERANDS if f16 else 0) | (TERM_PRODUCT if term else 0),
                                        None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(),
                                        torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "pika_gemm_nt(M=%d,N=%d,K=%d)" % (M, N, K))
    return out


def gemm_nt(a, b, bias=None, relu=False, out=None, accumulate=False, precision=None):
    """out[M,N] = act(a[M,K] @ b[N,K]^T + bias)."""
    a_op, M, K = matrix(a)
    b_op, N, Kb = matrix(b)
    if Kb != K:  # a zero-padded (bf16, transposed) operand may be wider than the fp32 one
        K = min(K, Kb)
        assert max(a.shape[1], b.shape[1]) - K < 8, (a.shape, b.shape)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert out.stride(1) == 1
    return launch(a_op, b_op, out, out.stride(0), M, N, K, bias, relu, accumulate, precision)


def gemm_bf16_nt(a, b, bias=None, out=None):
    """out[M,N] f32 = a[M,K] bf16 @ b[N,K]^T bf16 + bias through the direct-to-LDS kernel
    (include/pika_gemm.h: pika_gemm_bf16_nt).  K must be a multiple of 64."""
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    if not a.is_cuda:
        raise RuntimeError("pika_amd.gemm: tensors must live on a HIP device (no CPU path)")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert out.stride(1) == 1 and out.dtype == torch.float32
    with torch.cuda.device(a.device):
        rc = _lib.lib().pika_gemm_bf16_nt(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                                          out.data_ptr(), out.stride(0), M, N, K,
                                          None if bias is None else bias.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "pika_gemm_bf16_nt(M=%d,N=%d,K=%d)" % (M, N, K))
    return out


# ---- pika_gemm_bf16_ex: every epilogue of the direct-to-LDS kernel over an operand descriptor (two-term operands) ----
EPI_F32, EPI_DROPOUT_BF16, EPI_MASK_BF16, EPI_DROPOUT_RESIDUAL = 0, 1, 2, 3


class GemmEx(ctypes.Structure):
    _fields_ = [("A", Operand), ("B", ctypes.c_void_p), ("ldb", ctypes.c_longlong), ("M", ctypes.c_int),
                ("N", ctypes.c_int), ("K", ctypes.c_int), ("bias", ctypes.c_void_p), ("relu", ctypes.c_int),
                ("epilogue", ctypes.c_int), ("out", ctypes.c_void_p), ("out_lo", ctypes.c_void_p),
                ("ldo", ctypes.c_longlong), ("p_drop", ctypes.c_float), ("seed", ctypes.c_uint),
                ("aux", ctypes.c_void_p), ("ld_aux", ctypes.c_longlong), ("scale", ctypes.c_float),
                ("residual", ctypes.c_void_p), ("ld_res", ctypes.c_longlong)]


def gemm_ex(a_op, b, M, N, K, epilogue, out, out_lo=None, bias=None, relu=False, p_drop=0.0, seed=0, aux=None,
            scale=1.0, residual=None):
    """out = epilogue(A b^T): `a_op` an Operand (bf16; two-term with seg / lo_off), `b` a bf16 (N, >= K) matrix."""
    if not out.is_cuda:
        raise RuntimeError("pika_amd.gemm: tensors must live on a HIP device (no CPU path)")
    assert b.dtype == torch.bfloat16 and b.dim() == 2 and b.stride(1) == 1 and b.shape[1] >= K and b.shape[0] >= N
    assert out.stride(-1) == 1
    g = GemmEx()
    g.A = a_op
    g.B, g.ldb, g.M, g.N, g.K = b.data_ptr(), b.stride(0), M, N, K
    g.bias = None if bias is None else bias.data_ptr()
    g.relu, g.epilogue = int(bool(relu)), epilogue
    g.out, g.ldo = out.data_ptr(), out.stride(-2)
    if out_lo is not None:
        assert out_lo.stride(-2) == out.stride(-2) and out_lo.dtype == out.dtype == torch.bfloat16
        g.out_lo = out_lo.data_ptr()
    g.p_drop, g.seed = float(p_drop), int(seed)
    if aux is not None:
        g.aux, g.ld_aux = aux.data_ptr(), aux.stride(-2)
    g.scale = float(scale)
    if residual is not None:
        g.residual, g.ld_res = residual.data_ptr(), residual.stride(-2)
    with torch.cuda.device(out.device):
        rc = _lib.lib().pika_gemm_bf16_ex(ctypes.byref(g), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "pika_gemm_bf16_ex(M=%d,N=%d,K=%d,epi=%d)" % (M, N, K, epilogue))
    return out


def split_pair(x2d, Cp=None):
    """(hi, lo) bf16 planes (rows, Cp) of an f32 (rows, C) matrix (C % 8 == 0; columns [C, Cp) zero): x = hi + lo to 16
    mantissa bits.  The planes share one buffer (lo directly behind hi)."""
    assert x2d.dim() == 2 and x2d.dtype == torch.float32 and x2d.stride(1) == 1 and x2d.shape[1] % 8 == 0
    rows, C = x2d.shape
    Cp = _pad64(C) if Cp is None else Cp
    buf = torch.empty((2, rows, Cp), dtype=torch.bfloat16, device=x2d.device)
    with torch.cuda.device(x2d.device):
        rc = _lib.lib().pika_split_bf16_terms(x2d.data_ptr(), 1, rows, C, 0, x2d.stride(0), 0, 2, 2, Cp, buf.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "pika_split_bf16_terms(pair)")
    return buf[0], buf[1]


def split_weight(w2d, taps=1):
    """The B side of a two-term product: a bf16 (N, taps * 3 * Cp) copy of the f32 weight (N, taps * C) whose rows hold
    [hi | hi | lo] per tap, Cp = C padded to a multiple of 64."""
    N, K = w2d.shape
    C = K // taps
    assert C * taps == K and C % 8 == 0
    w = w2d.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    Cp = _pad64(C)
    op = Operand(w.data_ptr(), PIKA_F32, N * taps, N * taps, 0, C, C, 1, 0, 0, 0, 0)
    with torch.cuda.device(w.device):
        t = _split(op, 1, N * taps, C, 0, C, 1, 0, Cp, w.device, 2)
    return t.view(N, taps * 3 * Cp)


def pair_operand(hi, lo, rows, Cp, taps=1, dil=1, stride=1, pad=0, rows_per_batch=None, t_in=None, batch_stride=0):
    """Operand descriptor of the two-term A side over the planes (hi, lo), each (.., Cp) with row pitch hi.stride(-2)."""
    ld = hi.stride(-2)
    lo_off = (lo.data_ptr() - hi.data_ptr()) // 2
    assert lo.stride(-2) == ld and Cp % 64 == 0 and lo_off % 8 == 0
    rpb = rows if rows_per_batch is None else rows_per_batch
    op = Operand(hi.data_ptr(), PIKA_BF16, rpb, rpb if t_in is None else t_in, batch_stride, ld, 3 * Cp, stride, dil, pad,
                 0, 0)
    op.seg, op.lo_off = Cp, lo_off
    return op
