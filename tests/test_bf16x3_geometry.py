"""Host model of the "bf16x3" operand split (pika_amd.gemm._bf16x3_operands): the operand descriptors it builds
around the split copies must describe the SAME product (to the two-term accuracy) as the fp32 operands they
replace -- plain, time-delay (strided / dilated / padded) and reduction-major (`trans`) operands, per-tap segments,
padded segment widths, stacked batches.  Memory is emulated with numpy; the split kernel (ops.hip,
pika_split_bf16x3) by its definition in include/pika_ops.h; operand addressing by the definition in
include/pika_gemm.h.  The GPU tests (tests/test_gemm_gpu.py::test_bf16x3_*) run the same cases on the kernels."""
import numpy as np
import pytest
import torch

from pika_amd import gemm as G


class Mem(object):
    def __init__(self):
        self.bufs = {}
        self.next = 1 << 20

    def put(self, arr):
        ptr = self.next
        self.next += (arr.size * 8 + 4095) & ~4095
        self.bufs[ptr] = np.ascontiguousarray(arr, dtype=np.float64).reshape(-1)
        return ptr


class FakeTensor(object):
    def __init__(self, ptr):
        self._ptr = ptr

    def data_ptr(self):
        return self._ptr


def bf16(x):
    return torch.from_numpy(np.asarray(x, np.float32)).bfloat16().float().numpy().astype(np.float64)


def read(mem, op, r, k):
    """Element (row r, reduction k) of a non-`trans` operand (include/pika_gemm.h)."""
    b, t = divmod(r, op.rows_per_batch)
    tap, c = divmod(k, op.C)
    ti = t * op.stride + tap * op.dil - op.pad
    if ti < 0 or ti >= op.t_in:
        return 0.0
    return mem.bufs[op.ptr][b * op.batch_stride + ti * op.ld + c]


def dense(mem, op, extent, K):
    """(extent, K) matrix the operand stands for (output index, reduction index)."""
    if op.trans:
        return np.array([[read(mem, op, k, j) for k in range(K)] for j in range(extent)])
    return np.array([[read(mem, op, r, k) for k in range(K)] for r in range(extent)])


PATTERNS = {2: ((0, 1, 0), (0, 0, 1)), 3: ((0, 0, 1, 0, 2, 1), (0, 1, 0, 2, 0, 1))}   # include/pika_ops.h


def fake_split(mem):
    def split(op, n_batch, t_in, C, batch_stride, ld, role, layout, Cp, device, n_terms=2, pending=None):
        src = mem.bufs[op.ptr]
        x = np.zeros((n_batch, t_in, C))
        for b in range(n_batch):
            for t in range(t_in):
                off = b * batch_stride + t * ld
                x[b, t] = src[off: off + C]
        t0 = bf16(x)
        t1 = bf16(x - t0)
        t2 = bf16(x - t0 - t1)
        assert np.array_equal(t0 + t1 + t2, bf16(x) * 0 + np.asarray(x, np.float32).astype(np.float64))   # exact 8+8+8
        terms = (t0, t1, t2)
        segs = [terms[k] for k in PATTERNS[n_terms][role]]
        assert C % 8 == 0 and Cp % 8 == 0 and Cp >= C
        if layout == 0:
            dst = np.zeros((n_batch, t_in, len(segs), Cp))
            for s in range(len(segs)):
                dst[:, :, s, :C] = segs[s]
        else:
            assert Cp == C
            dst = np.stack(segs, 0)
        return FakeTensor(mem.put(dst))
    return split


def product(mem, a_op, b_op, M, N, K):
    return dense(mem, a_op, M, K) @ dense(mem, b_op, N, K).T


def check(monkeypatch, mem, a_op, b_op, M, N, K, expect_split=True):
    for n_terms in (2, 3):
        check_terms(monkeypatch, mem, a_op, b_op, M, N, K, expect_split, n_terms)


def check_terms(monkeypatch, mem, a_op, b_op, M, N, K, expect_split, n_terms):
    monkeypatch.setattr(G, "_split", fake_split(mem))
    want = product(mem, a_op, b_op, M, N, K)
    scale = np.abs(dense(mem, a_op, M, K)) @ np.abs(dense(mem, b_op, N, K)).T
    sp = G._bf16x3_operands(a_op, b_op, M, N, K, None, n_terms)
    if not expect_split:
        assert sp is None
        return
    assert sp is not None
    a3, b3, K3, keep = sp
    assert a3.dtype == b3.dtype == G.PIKA_BF16 and bool(a3.trans) == bool(a_op.trans) == bool(b3.trans)
    assert a_op.trans or K3 % 8 == 0       # the contiguous reduction of 16-byte bf16 loads
    got = product(mem, a3, b3, M, N, K3)
    err = np.abs(got - want) / np.maximum(scale, 1e-30)
    # a wrong role pairing (lo.lo, or hi.hi twice) misses these by orders of magnitude.  Three terms: the dropped
    # products (m.l, l.m, l.l) are below 2^-24 of the leading one
    assert err.max() < (4e-5 if n_terms == 2 else 3e-7), err.max()


def plain(mem, rng, rows, K, ld=None):
    ld = ld or K
    buf = rng.standard_normal((rows, ld)).astype(np.float32)
    return G.Operand(mem.put(buf), G.PIKA_F32, rows, rows, 0, ld, K, 1, 0, 0, 0, 0)


@pytest.mark.parametrize("M,N,K,lda", [(5, 7, 64, None), (9, 4, 24, 40), (3, 3, 200, None), (6, 5, 8, None)])
def test_plain_operands(monkeypatch, M, N, K, lda):
    mem, rng = Mem(), np.random.default_rng(M * 100 + K)
    check(monkeypatch, mem, plain(mem, rng, M, K, lda), plain(mem, rng, N, K), M, N, K)


@pytest.mark.parametrize("taps,dil,stride,pad,Bn,T,C", [(3, 1, 1, 0, 2, 9, 8), (3, 3, 1, 0, 2, 11, 16), (3, 2, 4, 0, 3, 14, 8),
                                                       (5, 1, 1, 4, 2, 7, 8), (2, 1, 2, 0, 1, 9, 24)])
def test_time_delay_forward_and_weight_gradient(monkeypatch, taps, dil, stride, pad, Bn, T, C):
    mem, rng = Mem(), np.random.default_rng(taps * 10 + T)
    N = 16
    x = rng.standard_normal((Bn, T, C)).astype(np.float32)
    t_out = (T + pad - dil * (taps - 1) - 1) // stride + 1 if pad == 0 else T
    M, K = Bn * t_out, taps * C
    a_op = G.Operand(mem.put(x), G.PIKA_F32, t_out, T, T * C, C, C, stride, dil, pad, 0, 0)
    w_op = plain(mem, rng, N, K)
    check(monkeypatch, mem, a_op, w_op, M, N, K)
    check(monkeypatch, mem, w_op, a_op, N, M, K)          # the time-delay view on the B side
    if pad == 0:
        # dW[n, (tap,c)] = sum_(b,t) dY[(b,t), n] * X[(b,t), (tap,c)]: both operands reduction-major
        dy_op = plain(mem, rng, M, N)
        dy_op.trans = 1
        a_t = G.Operand(a_op.ptr, G.PIKA_F32, t_out, T, T * C, C, C, stride, dil, 0, 0, 0)
        a_t.trans = 1
        check(monkeypatch, mem, dy_op, a_t, N, K, M)


def test_plain_reduction_major_and_refusals(monkeypatch):
    mem, rng = Mem(), np.random.default_rng(5)
    K, M, N = 13, 16, 8
    at, bt = plain(mem, rng, K, M), plain(mem, rng, K, N)
    at.trans = bt.trans = 1
    check(monkeypatch, mem, at, bt, M, N, K)
    # mixed orientation, bf16 operands, batched (z) operands, extents the 16-byte bf16 loads cannot take: not split
    a = plain(mem, rng, M, 16)
    check(monkeypatch, mem, a, bt, M, N, 13, expect_split=False)
    b16 = plain(mem, rng, N, 16)
    b16.dtype = G.PIKA_BF16
    check(monkeypatch, mem, a, b16, M, N, 16, expect_split=False)
    z = plain(mem, rng, N, 16)
    z.z_inner = 16
    check(monkeypatch, mem, a, z, M, N, 16, expect_split=False)
    check(monkeypatch, mem, plain(mem, rng, M, 12), plain(mem, rng, N, 12), M, N, 12, expect_split=False)
    odd = plain(mem, rng, K, 12)
    odd.trans = 1
    check(monkeypatch, mem, odd, bt, 12, N, K, expect_split=False)
