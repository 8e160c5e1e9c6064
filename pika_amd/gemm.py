"""Host wrapper of the MFMA GEMM (include/pika_gemm.h): builds operand descriptors from torch
tensors (device memory + strides only) and launches on torch's current stream."""
import ctypes
import os

import torch

from . import _lib

PIKA_F32, PIKA_BF16 = 0, 1
RELU, ACCUMULATE, FP32SPLIT = 1, 2, 4

# "bf16": one MFMA per product (config-2 arithmetic).  "fp32": hi/lo split, ~fp32 accuracy
# (parity runs).  Overridable per call.
PRECISION = os.environ.get("PIKA_GEMM_PRECISION", "bf16")


class Operand(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("dtype", ctypes.c_int), ("rows_per_batch", ctypes.c_int),
                ("t_in", ctypes.c_int), ("batch_stride", ctypes.c_longlong), ("ld", ctypes.c_longlong),
                ("C", ctypes.c_int), ("stride", ctypes.c_int), ("dil", ctypes.c_int),
                ("pad", ctypes.c_int), ("z_outer", ctypes.c_longlong), ("z_inner", ctypes.c_longlong),
                ("trans", ctypes.c_int)]


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
    if p not in ("bf16", "fp32"):
        raise ValueError("unknown GEMM precision %r" % (p,))
    return (RELU if relu else 0) | (ACCUMULATE if accumulate else 0) | (FP32SPLIT if p == "fp32" else 0)


OUT_BF16 = 8


_WORKSPACE = {}
WORKSPACE_BYTES = 160 << 20


def _workspace(device):
    """Per-device scratch for split-K partial tiles (stream-ordered use on the current stream)."""
    ws = _WORKSPACE.get(device.index)
    if ws is None:
        ws = _WORKSPACE[device.index] = torch.empty(WORKSPACE_BYTES, dtype=torch.uint8, device=device)
    return ws


def launch(a_op, b_op, out, ldc, M, N, K, bias=None, relu=False, accumulate=False, precision=None,
           batch=1, z_div=1, c_z_outer=0, c_z_inner=0):
    if not out.is_cuda:
        raise RuntimeError("pika_amd.gemm: tensors must live on a HIP device (no CPU path)")
    with torch.cuda.device(out.device):
        ws = _workspace(out.device) if (a_op.trans and b_op.trans) else None
        rc = _lib.lib().pika_gemm_nt_ws(ctypes.byref(a_op), ctypes.byref(b_op), out.data_ptr(), ldc,
                                        c_z_outer, c_z_inner, M, N, K, batch, z_div,
                                        None if bias is None else bias.data_ptr(),
                                        _flags(relu, accumulate, precision) | (OUT_BF16 if out.dtype == torch.bfloat16 else 0),
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
