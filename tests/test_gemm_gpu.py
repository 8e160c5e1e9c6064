"""MFMA GEMM (include/pika_gemm.h) vs fp64 matmul of the same operands.
bf16 mode: products of bf16-rounded operands, fp32 accumulate -> compared against the fp64
product OF THE ROUNDED operands (tight) and of the raw operands (loose, 2^-8 relative per
element).  fp32 mode (3-way bf16 split, 6 MFMAs): fp32-class accuracy -> 1e-6 of the row scale."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def ref(a, b, bias=None, relu=False):
    c = a.double() @ b.double().t()
    if bias is not None:
        c = c + bias.double()
    return torch.relu(c) if relu else c


def err_scale(a, b):
    return (a.double().abs() @ b.double().abs().t()).clamp_min(1e-30)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (16, 16, 32), (1, 4, 4), (300, 200, 64),
                                   (257, 130, 100), (1000, 5000, 1024), (2048, 1024, 240),
                                   (333, 100, 3072), (200, 300, 8192), (1024, 1024, 32000)])  # last two: split-K path
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_plain_nt(hip_device, M, N, K, precision):
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    # asymmetric, non-square data: a transposed or row/col-swapped kernel cannot pass
    a = (torch.randn(M, K, generator=g) + 0.3 * torch.arange(K) / K).to(hip_device)
    b = (torch.randn(N, K, generator=g) * (1 + torch.arange(N).unsqueeze(1) / N)).to(hip_device)
    bias = torch.randn(N, generator=g).to(hip_device)
    out = G.gemm_nt(a, b, bias=bias, relu=False, precision=precision)
    torch.cuda.synchronize()
    if precision == "bf16":
        ar, br = a.bfloat16().float(), b.bfloat16().float()
        e = (out.double() - ref(ar, br, bias)).abs() / err_scale(ar, br)
        assert e.max().item() < 2e-6 * max(1.0, K ** 0.5 / 8), e.max().item()  # fp32 accumulation only
        e2 = (out.double() - ref(a, b, bias)).abs() / err_scale(a, b)
        assert e2.max().item() < 2 ** -7
    else:
        e = (out.double() - ref(a, b, bias)).abs() / err_scale(a, b)
        assert e.max().item() < 1e-6 * max(1.0, K ** 0.5 / 8), e.max().item()


def test_epilogues_and_strided_output(hip_device):
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(5)
    a = torch.randn(200, 96, generator=g).to(hip_device)
    b = torch.randn(72, 96, generator=g).to(hip_device)
    bias = torch.randn(72, generator=g).to(hip_device)
    big = torch.full((200, 160), 7.0, device=hip_device)
    out = big[:, 40:112]  # ldc = 160, column offset: 16-byte aligned
    G.gemm_nt(a, b, bias=bias, relu=True, out=out, precision="fp32")
    want = ref(a, b, bias, relu=True)
    assert ((out.double() - want).abs() / err_scale(a, b).clamp_min(1)).max() < 2e-6
    assert torch.all(big[:, :40] == 7.0) and torch.all(big[:, 112:] == 7.0)
    prev = out.clone()
    G.gemm_nt(a, b, out=out, accumulate=True, precision="fp32")
    assert ((out.double() - (prev.double() + ref(a, b))).abs() / err_scale(a, b).clamp_min(1)).max() < 2e-6
    # unaligned output rows (ldc % 4 != 0) take the scalar epilogue
    odd = torch.zeros(200, 73, device=hip_device)
    G.gemm_nt(a, b, bias=bias, out=odd[:, :72], precision="fp32")
    assert ((odd[:, :72].double() - ref(a, b, bias)).abs() / err_scale(a, b).clamp_min(1)).max() < 2e-6


def test_bf16_operands_and_errors(hip_device):
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(6)
    a = torch.randn(130, 64, generator=g).to(hip_device)
    b = torch.randn(90, 64, generator=g).to(hip_device)
    o1 = G.gemm_nt(a.bfloat16(), b.bfloat16(), precision="bf16")
    o2 = G.gemm_nt(a, b, precision="bf16")
    assert torch.equal(o1, o2)  # rounding on the fly == pre-rounded operands
    o3 = G.gemm_nt(a.bfloat16(), b, precision="bf16")
    assert torch.equal(o1, o3)
    with pytest.raises(RuntimeError):  # K % 4 != 0 is refused, not silently mis-computed
        G.gemm_nt(a[:, :63].contiguous(), b[:, :63].contiguous())


@pytest.mark.parametrize("taps,dil,stride,pad,T,C,N", [(3, 1, 1, 0, 50, 64, 96), (3, 3, 1, 0, 77, 32, 40),
                                                     (3, 3, 4, 0, 90, 64, 64), (5, 1, 1, 4, 23, 16, 48)])
def test_time_delay_operand(hip_device, taps, dil, stride, pad, T, C, N):
    """Virtual im2col operand == explicit shifted-view concatenation (model/ops.py tdnn /
    causal_conv1d, which mirror rnnt_tdnn_transformer.py:44-57 and rnnt_conv_transformer_lm.py:73)."""
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(taps * 100 + dil * 10 + stride)
    Bn = 3
    x = torch.randn(Bn, T, C, generator=g).to(hip_device)
    w = torch.randn(N, taps * C, generator=g).to(hip_device)
    a_op, M, K, t_out = G.time_delay(x, taps, dil, stride, pad)
    b_op, _, _ = G.matrix(w)
    out = torch.empty(M, N, device=hip_device)
    G.launch(a_op, b_op, out, N, M, N, K, precision="fp32")
    xp = torch.nn.functional.pad(x, (0, 0, pad, 0))
    cols = [xp[:, j * dil: j * dil + (t_out - 1) * stride + 1: stride, :] for j in range(taps)]
    a = torch.cat(cols, -1).reshape(M, K)
    want = ref(a, w)
    assert ((out.double() - want).abs() / err_scale(a, w)).max() < 2e-6


def test_batched_strided_attention_shapes(hip_device):
    """scores[b,h] = q[b,:,h,:] @ k[b,:,h,:]^T on (B,T,H*D) tensors: z = b*H + h."""
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(9)
    Bn, T, H, D = 2, 70, 4, 16
    q = torch.randn(Bn, T, H * D, generator=g).to(hip_device)
    k = torch.randn(Bn, T, H * D, generator=g).to(hip_device)
    a_op = G.Operand(q.data_ptr(), 0, T, T, 0, H * D, D, 1, 0, 0, T * H * D, D)
    b_op = G.Operand(k.data_ptr(), 0, T, T, 0, H * D, D, 1, 0, 0, T * H * D, D)
    out = torch.empty(Bn, H, T, T, device=hip_device)
    G.launch(a_op, b_op, out, T, T, T, D, precision="fp32", batch=Bn * H, z_div=H,
             c_z_outer=H * T * T, c_z_inner=T * T)
    qh = q.view(Bn, T, H, D).transpose(1, 2).double()
    kh = k.view(Bn, T, H, D).transpose(1, 2).double()
    want = qh @ kh.transpose(2, 3)
    scale = qh.abs() @ kh.abs().transpose(2, 3)
    assert ((out.double() - want).abs() / scale).max() < 2e-6


@pytest.mark.parametrize("M,N,K", [(64, 128, 40), (300, 256, 1000), (1024, 3072, 5000), (100, 36, 77)])
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_transposed_operands_tn(hip_device, M, N, K, precision):
    """C[M,N] = At^T @ Bt with At (K,M), Bt (K,N) read in place through `trans` operands
    (weight-gradient form dW = dY^T X), and the mixed form A (M,K) normal x Bt (K,N) trans
    (input-gradient form dX = dY W)."""
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    at = (torch.randn(K, M, generator=g) + 0.2 * torch.arange(M) / M).to(hip_device)
    bt = (torch.randn(K, N, generator=g) * (1 + torch.arange(K).unsqueeze(1) / K)).to(hip_device)
    a_op, m_, k_ = G.matrix_t(at)
    b_op, n_, _ = G.matrix_t(bt)
    assert (m_, n_, k_) == (M, N, K)
    out = torch.empty(M, N, device=hip_device)
    G.launch(a_op, b_op, out, N, M, N, K, precision=precision)
    want = at.double().t() @ bt.double()
    scale = at.double().abs().t() @ bt.double().abs()
    tol = 1e-6 * max(1.0, K ** 0.5 / 8) if precision == "fp32" else 2 ** -7
    assert ((out.double() - want).abs() / scale).max().item() < tol
    if K % 4 == 0:
        a = at.t().contiguous()                      # (M,K) normal operand
        out2 = torch.empty(M, N, device=hip_device)
        G.launch(G.matrix(a)[0], G.matrix_t(bt)[0], out2, N, M, N, K, precision=precision)
        assert ((out2.double() - want).abs() / scale).max().item() < tol
        if precision == "bf16":   # bf16 storage of the transposed operand (joint hidden h)
            out3 = torch.empty(M, N, device=hip_device)
            if N % 8 == 0:
                G.launch(G.matrix(a)[0], G.matrix_t(bt.bfloat16())[0], out3, N, M, N, K, precision=precision)
                assert torch.allclose(out3, out2, rtol=1e-4, atol=1e-2)  # split-K sums in any order


def test_time_delay_weight_gradient_in_place(hip_device):
    """dW[n,(tap,c)] = sum_(b,t) dY[(b,t),n] * x[b, t*stride + tap*dil - pad, c] with both operands
    `trans` (one of them virtual) vs the explicit im2col product."""
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(77)
    Bn, T, C, N, taps, dil, stride, pad = 3, 61, 32, 48, 3, 3, 4, 0
    x = torch.randn(Bn, T, C, generator=g).to(hip_device)
    a_op, M, K, t_out = G.time_delay(x, taps, dil, stride, pad)
    dy = torch.randn(M, N, generator=g).to(hip_device)
    a_op.trans = 1
    dy_op = G.matrix(dy)[0]
    dy_op.trans = 1
    out = torch.empty(N, K, device=hip_device)
    G.launch(dy_op, a_op, out, K, N, K, M, precision="fp32")
    cols = [x[:, j * dil: j * dil + (t_out - 1) * stride + 1: stride, :] for j in range(taps)]
    a = torch.cat(cols, -1).reshape(M, K)
    want = dy.double().t() @ a.double()
    assert ((out.double() - want).abs() / (dy.double().abs().t() @ a.double().abs())).max() < 2e-6


@pytest.mark.parametrize("M,N,K", [(300, 200, 192), (1, 8, 64), (257, 132, 64), (1000, 5000, 1024),
                                   (777, 1024, 5056),
                                   # more output tiles than CUs: persistent workgroups walk several tiles each, the fetch
                                   # cursor crosses tile boundaries after 3 / 1 / 2 K-tiles, ragged edge tiles
                                   (4300, 4000, 192), (4300, 4000, 64), (2100, 8200, 128)])
def test_direct_to_lds_bf16_nt(hip_device, M, N, K):
    """pika_gemm_bf16_nt (gemm_glds.hip): operands are exactly representable bf16, so the only error
    against the fp64 product is fp32 accumulation."""
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16()
    b = torch.randn(N, K, generator=g).bfloat16()
    bias = torch.randn(N, generator=g)
    want = a.double() @ b.double().t() + bias.double()
    got = G.gemm_bf16_nt(a.to(hip_device), b.to(hip_device), bias=bias.to(hip_device))
    tol = 4e-6 * (a.double().abs() @ b.double().abs().t()).max().item() + 1e-6
    assert (got.double().cpu() - want).abs().max() < tol
    # padded leading dimensions + no bias, output into a strided view
    ap = torch.zeros(M, K + 64, dtype=torch.bfloat16, device=hip_device)
    ap[:, :K] = a.to(hip_device)
    Np = (N + 3) & ~3
    out = torch.full((M, Np + 4), -7.0, device=hip_device)
    G.gemm_bf16_nt(ap[:, :K], b.to(hip_device), out=out[:, :N])
    assert (out[:, :N].double().cpu() - (want - bias.double())).abs().max() < tol
    assert bool((out[:, N:] == -7.0).all())
    with pytest.raises(RuntimeError):
        G.gemm_bf16_nt(a.to(hip_device)[:, :K - 32], b.to(hip_device)[:, :K - 32])


@pytest.mark.parametrize("taps,dil,stride,T,C,N,B", [(3, 1, 1, 300, 64, 256, 3), (3, 3, 1, 200, 128, 200, 4),
                                                     (3, 3, 4, 410, 64, 512, 3), (1, 1, 1, 350, 192, 264, 2)])
def test_ping_pong_kernel_routes(hip_device, taps, dil, stride, T, C, N, B):
    """bf16 x bf16 operands (plain and time-delay views, bias, ReLU) take the 256x256 direct-to-LDS kernel
    inside pika_gemm_nt; PIKA_GEMM_NO_PP-free cross-check against the fp64 product."""
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(taps * 100 + T)
    x = torch.randn(B, T, C, generator=g).bfloat16()
    w = (torch.randn(N, taps * C, generator=g) * 0.1).bfloat16()
    bias = torch.randn(N, generator=g)
    t_out = (T - dil * (taps - 1) - 1) // stride + 1
    cols = torch.stack([x.double()[:, tap * dil: tap * dil + (t_out - 1) * stride + 1: stride] for tap in range(taps)], 2)
    want = torch.relu(cols.reshape(B * t_out, taps * C) @ w.double().t() + bias.double())
    xd, wd = x.to(hip_device), w.to(hip_device)
    a_op, M, K, t_o = G.time_delay(xd, taps, dil, stride, 0)
    assert (M, K, t_o) == (B * t_out, taps * C, t_out)
    out = torch.full((M, N), float("nan"), device=hip_device)
    G.launch(a_op, G.matrix(wd)[0], out, N, M, N, K, bias=bias.to(hip_device), relu=True)
    tol = 4e-6 * (cols.reshape(M, K).abs() @ w.double().abs().t()).max().item() + 1e-6
    assert (out.double().cpu() - want).abs().max() < tol


@pytest.mark.parametrize("Bn,T,C,N,taps,dil,stride,ldpad", [
    (4, 300, 256, 512, 3, 3, 1, 0),     # time-delay X, batch wrap inside K-tiles, ragged reduction
    (4, 310, 256, 512, 3, 3, 1, 0),     # the same with a reduction of 19 whole K-tiles (the no-zero-page instantiation)
    (5, 200, 256, 264, 1, 1, 1, 24),    # plain X, dY wider than its valid columns (zero page past the width)
    (2, 1100, 512, 1000, 3, 1, 4, 8),   # strided time-delay, output extents not multiples of 256
])
def test_ping_pong_weight_gradient(hip_device, Bn, T, C, N, taps, dil, stride, ldpad):
    """dW = dY^T X with both operands bf16 and reduction-major (`trans`): the 256x256 transpose-read
    kernel of gemm_glds.hip (split-K + atomics) vs the fp64 im2col product."""
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(Bn * 1000 + T)
    x = torch.randn(Bn, T, C, generator=g).bfloat16().to(hip_device)
    a_op, M, K, t_out = G.time_delay(x, taps, dil, stride, 0)
    dyp = torch.randn(M, N + ldpad, generator=g).bfloat16().to(hip_device)
    dy = dyp[:, :N]
    a_op.trans = 1
    dy_op = G.matrix(dy)[0]
    dy_op.trans = 1
    out = torch.full((N, K), float("nan"), device=hip_device)
    G.launch(dy_op, a_op, out, K, N, K, M)
    cols = [x[:, j * dil: j * dil + (t_out - 1) * stride + 1: stride, :] for j in range(taps)]
    a = torch.cat(cols, -1).reshape(M, K).double().cpu()
    want = dy.double().cpu().t() @ a
    scale = dy.double().cpu().abs().t() @ a.abs()
    assert ((out.double().cpu() - want).abs() / scale).max() < 1e-5   # exact bf16 products, fp32 sums


@pytest.mark.parametrize("taps,dil,stride,pad,Bn,T,C,N", [(3, 3, 1, 0, 8, 1100, 256, 256), (3, 1, 1, 0, 9, 1000, 256, 320),
                                                         (5, 1, 1, 4, 8, 1100, 256, 256), (3, 3, 4, 0, 8, 1100, 256, 256)])
def test_time_delay_fn_bf16_gradients(hip_device, taps, dil, stride, pad, Bn, T, C, N):
    """TimeDelayFn in the bf16 mode at sizes that take the direct-to-LDS kernels: forward (time-delay view,
    bias, ReLU), dX as ONE GEMM over the padded, tap-reversed view of dY (stride 1) and dW through the
    transposed-operand kernel, vs fp64 conv1d autograd on the bf16-rounded operands."""
    import torch.nn.functional as F
    from pika_amd import gemm as G
    from pika_amd.model.hipops import TimeDelayFn
    old, G.PRECISION = G.PRECISION, "bf16"
    try:
        g = torch.Generator().manual_seed(taps * 10 + dil)
        x = torch.randn(Bn, T, C, generator=g).bfloat16().float()
        w = (torch.randn(N, taps * C, generator=g) * 0.05).bfloat16().float()
        b = torch.randn(N, generator=g)
        xr = x.double().requires_grad_(True)
        wr = w.double().requires_grad_(True)
        br = b.double().requires_grad_(True)
        w3 = wr.view(N, taps, C).permute(0, 2, 1)                       # (N, C, taps)
        xin = F.pad(xr.transpose(1, 2), (pad, 0)) if pad else xr.transpose(1, 2)
        yr = F.conv1d(xin, w3, br, stride=stride, dilation=dil).transpose(1, 2)
        if pad:
            yr = yr[:, :T]
        gy = torch.randn(yr.shape, generator=g).bfloat16().float()
        (yr * gy.double()).sum().backward()
        xd = x.to(hip_device).requires_grad_(True)
        wd = w.to(hip_device).requires_grad_(True)
        bd = b.to(hip_device).requires_grad_(True)
        y = TimeDelayFn.apply(xd, wd, bd, taps, dil, stride, pad, 0)
        assert y.shape == yr.shape
        assert (y.double().cpu() - yr.detach()).abs().max() < 1e-4 * yr.detach().abs().max()
        (y * gy.to(hip_device)).sum().backward()
        for got, want in ((xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
            assert (got.double().cpu() - want).abs().max() < 1e-4 * want.abs().max()
    finally:
        G.PRECISION = old


@pytest.mark.parametrize("M,d,F,p", [(300, 64, 256, 0.0), (1000, 128, 512, 0.2), (513, 64, 192, 0.5)])
def test_feed_forward_fn(hip_device, M, d, F, p):
    """FeedForwardFn (bf16-only hidden, ReLU + dropout in the GEMM epilogues) vs the fp64 chain
    w_2(dropout(relu(w_1(x)))) with the kernel's own keep-mask, on bf16-representable inputs."""
    from pika_amd import gemm as G
    from pika_amd.model.hipops import FeedForwardFn, dropout_keep_mask
    old, G.PRECISION = G.PRECISION, "bf16"
    try:
        g = torch.Generator().manual_seed(M + F)
        x = torch.randn(M, d, generator=g).bfloat16().float()
        w1 = (torch.randn(F, d, generator=g) * 0.2).bfloat16().float()
        b1 = torch.randn(F, generator=g) * 0.1
        w2 = (torch.randn(d, F, generator=g) * 0.1).bfloat16().float()
        b2 = torch.randn(d, generator=g) * 0.1
        gy = torch.randn(M, d, generator=g)
        seed = 4242
        dev = [t.to(hip_device).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        y = FeedForwardFn.apply(*dev, p, seed)
        (y * gy.to(hip_device)).sum().backward()
        keep = dropout_keep_mask(M, F, p, seed, hip_device).cpu()
        thr = round(p * 65536)
        if p > 0:
            assert abs(keep.float().mean().item() - (1 - p)) < 4 * (p * (1 - p) / keep.numel()) ** 0.5 + 1e-4
            assert abs(keep.float().mean(0).std().item() - (p * (1 - p) / M) ** 0.5) < 0.3 * (p * (1 - p) / M) ** 0.5
        ref = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        h = torch.relu(ref[0] @ ref[1].t() + ref[2]) * keep.double() * (65536.0 / (65536 - thr))
        yr = h @ ref[3].t() + ref[4]
        (yr * gy.double()).sum().backward()
        # the hidden and dh are rounded to bf16 (2^-9 relative per element, unbiased)
        assert (y.double().cpu() - yr.detach()).abs().max() < 1e-2 * yr.detach().abs().max()
        for a, b in zip(dev, ref):
            s = b.grad.abs().max().item()
            assert (a.grad.double().cpu() - b.grad).abs().max() < 1.5e-2 * s, (a.shape, s)
            assert (a.grad.double().cpu() - b.grad).norm() < 6e-3 * b.grad.norm()
    finally:
        G.PRECISION = old


@pytest.mark.parametrize("M,K,N,p", [(300, 64, 128, 0.0), (1000, 128, 256, 0.2)])
def test_linear_dropout_residual_fn(hip_device, M, K, N, p):
    """LinearDropoutResidualFn: dropout(x W^T + b) + residual in one GEMM epilogue, backward with the mask
    re-applied while rounding the gradient to bf16; vs the fp64 formula with the kernel's own keep mask."""
    from pika_amd import gemm as G
    from pika_amd.model.hipops import LinearDropoutResidualFn, dropout_keep_mask
    old, G.PRECISION = G.PRECISION, "bf16"
    try:
        g = torch.Generator().manual_seed(M + N)
        x = torch.randn(M, K, generator=g).bfloat16()
        w = (torch.randn(N, K, generator=g) * 0.2).bfloat16().float()
        b = torch.randn(N, generator=g) * 0.1
        res = torch.randn(M, N, generator=g)
        gy = torch.randn(M, N, generator=g)
        seed = 777
        xd = x.to(hip_device).requires_grad_(True)
        wd, bd, rd = [t.to(hip_device).requires_grad_(True) for t in (w, b, res)]
        y = LinearDropoutResidualFn.apply(xd, wd, bd, rd, p, seed)
        (y * gy.to(hip_device)).sum().backward()
        keep = dropout_keep_mask(M, N, p, seed, hip_device).cpu().double()
        sc = 65536.0 / (65536 - round(p * 65536))
        xr, wr, br, rr = [t.double().requires_grad_(True) for t in (x.float(), w, b, res)]
        yr = (xr @ wr.t() + br) * keep * sc + rr
        (yr * gy.double()).sum().backward()
        assert (y.double().cpu() - yr.detach()).abs().max() < 1e-4 * yr.detach().abs().max()
        assert torch.equal(rd.grad.cpu(), gy)                     # the residual gets the gradient as is
        for got, want in ((xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
            s = want.abs().max().item()
            assert (got.double().cpu() - want).abs().max() < 1.5e-2 * s
            assert (got.double().cpu() - want).norm() < 6e-3 * want.norm()
    finally:
        G.PRECISION = old


@pytest.mark.parametrize("taps,dil,stride,out_bf16,x_bf16", [(3, 3, 1, False, False), (3, 1, 1, True, True),
                                                             (3, 3, 4, False, True)])
def test_tdnn_bn_fn(hip_device, taps, dil, stride, out_bf16, x_bf16):
    """TdnnBnFn = bn(relu(time_delay(x))) as one node (bf16 dy inside, optional bf16 in/out) vs fp64 torch
    (conv1d -> relu -> batch_norm in training mode) on bf16-representable inputs."""
    import torch.nn.functional as F
    from pika_amd import gemm as G
    from pika_amd.model.hipops import TdnnBnFn
    old, G.PRECISION = G.PRECISION, "bf16"
    try:
        g = torch.Generator().manual_seed(taps * 7 + stride)
        Bn, T, C, N = 8, 1100, 256, 256
        x = torch.randn(Bn, T, C, generator=g).bfloat16().float()
        w = (torch.randn(N, taps * C, generator=g) * 0.05).bfloat16().float()
        b = torch.randn(N, generator=g) * 0.1
        gam = torch.rand(N, generator=g) + 0.5
        bet = torch.randn(N, generator=g) * 0.1
        ref = [t.double().requires_grad_(True) for t in (x, w, b, gam, bet)]
        w3 = ref[1].view(N, taps, C).permute(0, 2, 1)
        yr = torch.relu(F.conv1d(ref[0].transpose(1, 2), w3, ref[2], stride=stride, dilation=dil)).transpose(1, 2)
        t_out = yr.shape[1]
        outr = F.batch_norm(yr.reshape(-1, N), None, None, ref[3], ref[4], True, 0.1, 1e-5).view(Bn, t_out, N)
        gy = torch.randn(outr.shape, generator=g).bfloat16().float()
        (outr * gy.double()).sum().backward()
        xd = (x.bfloat16() if x_bf16 else x).to(hip_device).requires_grad_(True)
        dev = [t.to(hip_device).requires_grad_(True) for t in (w, b, gam, bet)]
        rm, rv = torch.zeros(N, device=hip_device), torch.ones(N, device=hip_device)
        out = TdnnBnFn.apply(xd, dev[0], dev[1], taps, dil, stride, 0, dev[2], dev[3], rm, rv, 1e-5, 0.1, out_bf16)
        assert out.dtype == (torch.bfloat16 if out_bf16 else torch.float32) and out.shape == outr.shape
        tol = 2 ** -7 if out_bf16 else 2e-4
        assert (out.double().cpu() - outr.detach()).abs().max() < tol * outr.detach().abs().max()
        (out * gy.to(hip_device).to(out.dtype)).sum().backward()
        # running statistics as nn.BatchNorm1d updates them
        assert torch.allclose(rm.double().cpu(), 0.1 * yr.detach().reshape(-1, N).mean(0), atol=1e-4)
        for got, want in zip([xd] + dev, ref):
            s = want.grad.abs().max().item()
            d = (got.grad.double().cpu() - want.grad)
            assert d.abs().max() < 3e-2 * s, (tuple(got.shape), d.abs().max().item(), s)   # bf16 dy (2^-9 per element)
            assert d.norm() < 1e-2 * want.grad.norm()
    finally:
        G.PRECISION = old


def test_bf16_output_flag_on_the_generic_entry(hip_device):
    """PIKA_GEMM_OUT_BF16 through pika_gemm_nt (G.launch with a bf16 `out`): plain and padded time-delay A operands
    at a size the direct-to-LDS kernel takes (>= 160 output tiles); equals the fp32 result rounded to bf16.  A
    product the kernel does not take must be refused, not silently written as fp32."""
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(8)
    M, N, K = 10240, 1024, 256
    a = torch.randn(M, K, generator=g).bfloat16().to(hip_device)
    b = (torch.randn(N, K, generator=g) * 0.1).bfloat16().to(hip_device)
    bias = torch.randn(N, generator=g).to(hip_device)
    ref = torch.empty(M, N, device=hip_device)
    G.launch(G.matrix(a)[0], G.matrix(b)[0], ref, N, M, N, K, bias=bias, relu=True)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=hip_device)
    G.launch(G.matrix(a)[0], G.matrix(b)[0], out, N, M, N, K, bias=bias, relu=True)
    assert torch.equal(out, ref.bfloat16())
    # the transposed convolution of TdnnBnFn.backward: padded, tap-reversed time-delay view of dY, bf16 dx
    Bn, T, C, taps, dil = 8, 1290, 1024, 3, 3
    t_out = T - dil * (taps - 1)
    dy = torch.randn(Bn * t_out, 256, generator=g).bfloat16().to(hip_device)
    wrev = (torch.randn(C, taps * 256, generator=g) * 0.05).bfloat16().to(hip_device)
    a_op = G.Operand(dy.data_ptr(), G.PIKA_BF16, T, t_out, t_out * 256, 256, 256, 1, dil, (taps - 1) * dil, 0, 0)
    dx32 = torch.empty(Bn * T, C, device=hip_device)
    G.launch(a_op, G.matrix(wrev)[0], dx32, C, Bn * T, C, taps * 256)
    dx16 = torch.empty(Bn * T, C, dtype=torch.bfloat16, device=hip_device)
    G.launch(a_op, G.matrix(wrev)[0], dx16, C, Bn * T, C, taps * 256)
    assert torch.equal(dx16, dx32.bfloat16())
    small = torch.empty(300, 200, dtype=torch.bfloat16, device=hip_device)
    with pytest.raises(RuntimeError):
        G.launch(G.matrix(a[:300])[0], G.matrix(b[:200])[0], small, 200, 300, 200, K)


# ---- K-concatenated term products (pika_split_bf16_terms): "bf16x3" = two bf16 terms per fp32 operand, hi.hi + lo.hi +
# hi.lo as one bf16 product; "fp32" at direct-to-LDS sizes = three terms, six segments ----

X3_TOL = 4e-5    # 2^-17 per operand + the dropped lo.lo term (2^-18), relative to sum |a||b|; fp32 accumulation on top


def _x3_tol(K):
    return X3_TOL + 2e-6 * max(1.0, K ** 0.5 / 8)


def test_split_terms_kernel_layouts(hip_device):
    """pika_split_bf16_terms: t0 = bf16(x), t1 = bf16(x - t0), t2 = bf16(x - t0 - t1) (exact), two / three terms, both
    layouts, both roles, zero pad columns, batched source with a pitch and a batch stride."""
    from pika_amd import _lib
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(3)
    nb, t_in, C, ld, Cp = 3, 7, 24, 40, 64
    src = torch.randn(nb, t_in + 2, ld, generator=g).to(hip_device)        # batch stride (t_in+2)*ld, pitch ld
    x = src[:, :t_in, :C]
    t0 = x.bfloat16()
    t1 = (x - t0.float()).bfloat16()
    t2 = (x - t0.float() - t1.float()).bfloat16()
    assert torch.equal(t0.float() + t1.float() + t2.float(), x)           # 8 + 8 + 8 mantissa bits: exact
    terms = (t0, t1, t2)
    patterns = {2: ((0, 1, 0), (0, 0, 1)), 3: ((0, 0, 1, 0, 2, 1), (0, 1, 0, 2, 0, 1))}    # include/pika_ops.h
    op = G.Operand(src.data_ptr(), G.PIKA_F32, t_in, t_in, (t_in + 2) * ld, ld, C, 1, 0, 0, 0, 0)
    for n_terms in (2, 3):
        for role in (0, 1):
            segs = [terms[k] for k in patterns[n_terms][role]]
            S = len(segs)
            cat = G._split(op, nb, t_in, C, (t_in + 2) * ld, ld, role, 0, Cp, hip_device, n_terms).view(nb, t_in, S, Cp)
            stk = G._split(op, nb, t_in, C, (t_in + 2) * ld, ld, role, 1, C, hip_device, n_terms).view(S, nb, t_in, C)
            torch.cuda.synchronize()
            for s_ in range(S):
                assert torch.equal(cat[:, :, s_, :C], segs[s_]) and bool((cat[:, :, s_, C:] == 0).all())
                assert torch.equal(stk[s_], segs[s_])
    with pytest.raises(RuntimeError):      # C % 8 != 0 is refused
        G._split(op, nb, t_in, 20, (t_in + 2) * ld, ld, 0, 0, 64, hip_device)
    with pytest.raises(RuntimeError):      # the stacked layout has no pad columns
        _lib.check(_lib.lib().pika_split_bf16_terms(src.data_ptr(), nb, t_in, C, (t_in + 2) * ld, ld, 0, 2, 1, Cp,
                                                    src.data_ptr(), None), "pika_split_bf16_terms")


@pytest.mark.parametrize("M,N,K", [(4096, 2560, 1024), (10240, 1024, 5000), (41000, 1024, 240)])
def test_fp32_mode_takes_the_six_segment_path_at_direct_to_lds_sizes(hip_device, M, N, K, monkeypatch):
    """Exact mode, products large enough for the direct-to-LDS kernels: three terms per operand, six segments, ONE bf16
    product -- the same six products PIKA_GEMM_FP32SPLIT issues.  Same accuracy bound as the register-staged exact
    kernel, and the two agree with each other to fp32 accumulation noise."""
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) + 0.3 * torch.arange(K) / K).to(hip_device)
    b = (torch.randn(N, K, generator=g) * (1 + torch.arange(N).unsqueeze(1) / N)).to(hip_device)
    bias = torch.randn(N, generator=g).to(hip_device)
    n0 = G.FP32_STATS["concat"]
    out = G.gemm_nt(a, b, bias=bias, relu=True, precision="fp32")
    assert G.FP32_STATS["concat"] == n0 + 1
    tol = 1e-6 * max(1.0, K ** 0.5 / 8)
    e = (out.double() - ref(a, b, bias, relu=True)).abs() / err_scale(a, b)
    assert e.max().item() < tol, e.max().item()
    monkeypatch.setattr(G, "FP32_CONCAT", False)
    s0 = G.FP32_STATS["concat"]
    staged = G.gemm_nt(a, b, bias=bias, relu=True, precision="fp32")
    assert G.FP32_STATS["concat"] == s0
    e2 = (staged.double() - out.double()).abs() / err_scale(a, b)
    assert e2.max().item() < tol


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (16, 16, 8), (300, 200, 64), (257, 136, 104), (1000, 5000, 1024),
                                   (2048, 1024, 240), (4096, 2560, 1024), (5120, 1024, 5000), (200, 304, 8192)])
def test_bf16x3_plain_nt(hip_device, M, N, K):
    """Plain products in the bf16x3 mode (incl. sizes the direct-to-LDS kernel takes, a reduction that is padded per
    segment: 240 -> 256, 5000 -> 5056) vs the fp64 product of the RAW fp32 operands."""
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = (torch.randn(M, K, generator=g) + 0.3 * torch.arange(K) / K).to(hip_device)
    b = (torch.randn(N, K, generator=g) * (1 + torch.arange(N).unsqueeze(1) / N)).to(hip_device)
    bias = torch.randn(N, generator=g).to(hip_device)
    n0 = G.BF16X3_STATS["fast"]
    out = G.gemm_nt(a, b, bias=bias, relu=True, precision="bf16x3")
    assert G.BF16X3_STATS["fast"] == n0 + 1            # took the split path, not the exact fallback
    e = (out.double() - ref(a, b, bias, relu=True)).abs() / err_scale(a, b)
    assert e.max().item() < _x3_tol(K), e.max().item()
    # and it IS more than one bf16 term: the one-term product misses the same bound by two orders of magnitude
    e1 = (G.gemm_nt(a, b, bias=bias, relu=True, precision="bf16").double() - ref(a, b, bias, relu=True)).abs() / err_scale(a, b)
    assert e1.max().item() > 20 * e.max().item()


def test_bf16x3_falls_back_to_the_exact_path(hip_device):
    """Operands the split cannot take (reduction not a multiple of 8, bf16 operands, batched products) run through
    the exact three-term kernel: same accuracy class or better, never an error."""
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(11)
    a = torch.randn(130, 100, generator=g).to(hip_device)
    b = torch.randn(90, 100, generator=g).to(hip_device)
    n0 = G.BF16X3_STATS["exact"]
    out = G.gemm_nt(a, b, precision="bf16x3")
    assert G.BF16X3_STATS["exact"] == n0 + 1
    assert ((out.double() - ref(a, b)).abs() / err_scale(a, b)).max() < 2e-6


@pytest.mark.parametrize("taps,dil,stride,pad,Bn,T,C,N", [(3, 1, 1, 0, 3, 50, 64, 96), (3, 3, 1, 0, 3, 77, 32, 40),
                                                         (3, 3, 4, 0, 3, 90, 64, 64), (5, 1, 1, 4, 3, 23, 16, 48),
                                                         (3, 3, 1, 0, 8, 1100, 256, 512), (5, 1, 1, 4, 8, 1100, 256, 256)])
def test_bf16x3_time_delay_forward_and_weight_gradient(hip_device, taps, dil, stride, pad, Bn, T, C, N):
    """Time-delay operands in the bf16x3 mode: forward (per-tap segments on both sides) and the weight gradient with
    both operands reduction-major (stacked segments; one of them the virtual view), vs fp64 im2col products."""
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(taps * 100 + dil * 10 + stride + T)
    x = torch.randn(Bn, T, C, generator=g).to(hip_device)
    w = torch.randn(N, taps * C, generator=g).to(hip_device)
    a_op, M, K, t_out = G.time_delay(x, taps, dil, stride, pad)
    out = torch.empty(M, N, device=hip_device)
    n0 = G.BF16X3_STATS["fast"]
    G.launch(a_op, G.matrix(w)[0], out, N, M, N, K, precision="bf16x3")
    xp = torch.nn.functional.pad(x, (0, 0, pad, 0))
    cols = [xp[:, j * dil: j * dil + (t_out - 1) * stride + 1: stride, :] for j in range(taps)]
    a = torch.cat(cols, -1).reshape(M, K)
    assert ((out.double() - ref(a, w)).abs() / err_scale(a, w)).max() < _x3_tol(K)
    assert G.BF16X3_STATS["fast"] == n0 + 1
    if pad == 0:
        dy = torch.randn(M, N, generator=g).to(hip_device)
        a_op = G.time_delay(x, taps, dil, stride, pad)[0]
        a_op.trans = 1
        dy_op = G.matrix(dy)[0]
        dy_op.trans = 1
        dw = torch.empty(N, K, device=hip_device)
        G.launch(dy_op, a_op, dw, K, N, K, M, precision="bf16x3")
        assert G.BF16X3_STATS["fast"] == n0 + 2
        want = dy.double().t() @ a.double()
        assert ((dw.double() - want).abs() / (dy.double().abs().t() @ a.double().abs())).max() < _x3_tol(M)


@pytest.mark.parametrize("M,N,K", [(64, 128, 40), (304, 256, 1000), (1024, 3072, 5000)])
def test_bf16x3_transposed_operands(hip_device, M, N, K):
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    at = (torch.randn(K, M, generator=g) + 0.2 * torch.arange(M) / M).to(hip_device)
    bt = (torch.randn(K, N, generator=g) * (1 + torch.arange(K).unsqueeze(1) / K)).to(hip_device)
    out = torch.empty(M, N, device=hip_device)
    n0 = G.BF16X3_STATS["fast"]
    G.launch(G.matrix_t(at)[0], G.matrix_t(bt)[0], out, N, M, N, K, precision="bf16x3")
    assert G.BF16X3_STATS["fast"] == n0 + 1
    want = at.double().t() @ bt.double()
    scale = at.double().abs().t() @ bt.double().abs()
    assert ((out.double() - want).abs() / scale).max().item() < _x3_tol(K)
    # mixed orientation (A normal, B reduction-major) is not split: exact path
    out2 = torch.empty(M, N, device=hip_device)
    G.launch(G.matrix(at.t().contiguous())[0], G.matrix_t(bt)[0], out2, N, M, N, K, precision="bf16x3")
    assert ((out2.double() - want).abs() / scale).max().item() < 2e-6 * max(1.0, K ** 0.5 / 8)
