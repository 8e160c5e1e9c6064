"""The kernels of the "mixed" arithmetic mode (two-term forward, bf16 backward: DESIGN 6.1) against fp64 formulas:
  * pika_gemm_bf16_ex over two-term operands (segment map [hi | lo | hi] x [hi | hi | lo]): plain, time-delay, padded
    views; every epilogue; the two-plane output
  * LayerNorm / BatchNorm two-plane outputs
  * the two-term attention forward (pika_attention_fwd_two_term), and the attention mask of the prediction network
    (multi_headed_attn.py:215-217) in all four attention kernels
Tolerances: a two-term product is exact to ~2^-16 of sum |a||b| (measured ~1e-5); the planes carry 16 mantissa bits."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def two_term(x):
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    return hi.double() + lo.double()


def pair_on(dev, x2d, Cp=None):
    from pika_amd import gemm as G
    return G.split_pair(x2d.to(dev).contiguous(), Cp)


@pytest.mark.parametrize("M,N,C", [(100, 64, 64), (2000, 1000, 240), (4096, 1024, 1024), (777, 3072, 512)])
def test_two_term_product_plain_every_epilogue(hip_device, M, N, C):
    from pika_amd import gemm as G
    from pika_amd.model.hipops import dropout_keep_mask
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, C, generator=g)
    w = torch.randn(N, C, generator=g) / math.sqrt(C)
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    hi, lo = pair_on(hip_device, x)
    Cp = hi.shape[1]
    assert Cp % 64 == 0 and Cp >= C
    # the planes: x = hi + lo to 2^-16, pad columns zero
    back = (hi.float() + lo.float())[:, :C].cpu()
    assert (back - x).abs().max() <= 2.0 ** -16 * x.abs().max()
    if Cp > C:
        assert not hi[:, C:].any() and not lo[:, C:].any()
    wd = w.to(hip_device)
    wb = G.split_weight(wd)
    assert wb.shape == (N, 3 * Cp)
    a_op = G.pair_operand(hi, lo, M, Cp)
    want = two_term(x) @ two_term(w).t() + b.double()
    scale = (x.abs().double() @ w.abs().double().t()).max().item()
    bd = b.to(hip_device)
    # f32 epilogue (+ bias, ReLU)
    out = torch.empty(M, N, device=hip_device)
    G.gemm_ex(a_op, wb, M, N, 3 * Cp, G.EPI_F32, out, bias=bd, relu=True)
    err = (out.double().cpu() - want.clamp(min=0)).abs().max().item()
    assert err < 4e-5 * scale, (err, scale)
    # ... which is far inside what ONE bf16 term gives
    one = (x.bfloat16().double() @ w.bfloat16().double().t() + b.double()).clamp(min=0)
    assert err < 0.05 * (one - want.clamp(min=0)).abs().max().item()
    # two-plane bf16 epilogue, with and without dropout
    for p in (0.0, 0.25):
        oh = torch.empty(M, N, dtype=torch.bfloat16, device=hip_device)
        ol = torch.empty_like(oh)
        G.gemm_ex(a_op, wb, M, N, 3 * Cp, G.EPI_DROPOUT_BF16, oh, out_lo=ol, bias=bd, relu=False, p_drop=p, seed=77)
        ref = want
        if p:
            keep = dropout_keep_mask(M, N, p, 77, hip_device).cpu()
            thr = round(p * 65536)
            ref = want * keep * (65536.0 / (65536 - thr))
            assert abs(keep.float().mean().item() - (1 - p)) < 0.01
        got = oh.double().cpu() + ol.double().cpu()
        assert (got - ref).abs().max().item() < 4e-5 * scale + 2.0 ** -15 * ref.abs().max().item()
        assert (ol.float().abs() <= 2.0 ** -8 * oh.float().abs() + 1e-30).all()     # lo is a rounding remainder of hi
    # dropout + residual epilogue
    out = torch.empty(M, N, device=hip_device)
    G.gemm_ex(a_op, wb, M, N, 3 * Cp, G.EPI_DROPOUT_RESIDUAL, out, bias=bd, p_drop=0.0, residual=res.to(hip_device))
    assert (out.double().cpu() - (want + res.double())).abs().max().item() < 4e-5 * scale


@pytest.mark.parametrize("taps,dil,stride,pad,C,T", [(3, 1, 1, 0, 64, 200), (3, 3, 1, 0, 256, 300), (3, 3, 4, 0, 128, 420),
                                                     (5, 1, 1, 4, 128, 51)])
def test_two_term_product_time_delay_views(hip_device, taps, dil, stride, pad, C, T):
    """TDNN layers (rnnt_tdnn_transformer.py:44-57) and the causal convolution (rnnt_conv_transformer_lm.py:36-45,73)
    as ONE two-term product over a time-delay view of the two planes."""
    from pika_amd import gemm as G
    Bn, N = 5, 192
    g = torch.Generator().manual_seed(taps * 100 + dil * 10 + stride)
    x = torch.randn(Bn, T, C, generator=g)
    w = torch.randn(N, taps * C, generator=g) / math.sqrt(taps * C)
    b = torch.randn(N, generator=g)
    hi, lo = pair_on(hip_device, x.view(-1, C))
    xd = x.to(hip_device)
    t_out = G.time_delay(xd, taps, dil, stride, pad)[3]
    a_op = G.pair_operand(hi, lo, Bn * t_out, C, taps, dil, stride, pad, rows_per_batch=t_out, t_in=T, batch_stride=T * C)
    out = torch.empty(Bn * t_out, N, device=hip_device)
    G.gemm_ex(a_op, G.split_weight(w.to(hip_device), taps), Bn * t_out, N, taps * 3 * C, G.EPI_F32, out, bias=b.to(hip_device))
    x2, w2 = two_term(x), two_term(w)
    xp = torch.nn.functional.pad(x2, (0, 0, pad, 0))
    cols = torch.cat([xp[:, j * dil: j * dil + (t_out - 1) * stride + 1: stride, :] for j in range(taps)], -1)
    want = cols @ w2.t() + b.double()
    scale = (cols.abs() @ w2.abs().t()).max().item()
    err = (out.view(Bn, t_out, N).double().cpu() - want).abs().max().item()
    assert err < 4e-5 * scale, (err, scale)


def test_norm_kernels_write_two_planes(hip_device):
    from pika_amd.model.hipops import BatchNormFn, LayerNormFn, Pair
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(300, 512, generator=g) * 3 + 1).to(hip_device)
    w, b = torch.randn(512, generator=g).to(hip_device), torch.randn(512, generator=g).to(hip_device)
    y32 = LayerNormFn.apply(x, w, b, 1e-6, False)
    hi, lo = LayerNormFn.apply(x, w, b, 1e-6, True, True)
    assert hi.dtype == lo.dtype == torch.bfloat16 and torch.equal(hi, y32.bfloat16())
    assert ((hi.float() + lo.float()) - y32).abs().max() <= 2.0 ** -16 * y32.abs().max()
    rm, rv = torch.zeros(512, device=hip_device), torch.ones(512, device=hip_device)
    z32 = BatchNormFn.apply(x, w, b, rm.clone(), rv.clone(), 1e-5, 0.1, False, False)
    zh, zl = BatchNormFn.apply(x, w, b, rm.clone(), rv.clone(), 1e-5, 0.1, False, True, True)
    assert torch.equal(zh, z32.bfloat16())
    assert ((zh.float() + zl.float()) - z32).abs().max() <= 2.0 ** -16 * z32.abs().max()
    # the planes share one buffer, lo directly behind hi: what gemm.pair_operand addresses
    assert zl.data_ptr() - zh.data_ptr() == zh.numel() * 2
    # gradients flow through the hi plane only
    xr = x.clone().requires_grad_(True)
    h2, l2 = LayerNormFn.apply(xr, w, b, 1e-6, True, True)
    assert not l2.requires_grad
    h2.float().sum().backward()
    assert xr.grad is not None and torch.isfinite(xr.grad).all()
    p = Pair(h2.detach(), l2)
    assert p.view(-1, 512).shape == (300, 512) and (p.float() - y32).abs().max() <= 2.0 ** -16 * y32.abs().max()


def attention_reference(q, k, v, heads, mask=None):
    B, T, HD = q.shape
    D = HD // heads
    qh = (q / math.sqrt(D)).view(B, T, heads, D).transpose(1, 2)
    kh = k.view(B, T, heads, D).transpose(1, 2)
    vh = v.view(B, T, heads, D).transpose(1, 2)
    s = qh @ kh.transpose(2, 3)
    if mask is not None:
        s = s.masked_fill(mask.unsqueeze(1), -1e18)
    return (torch.softmax(s, dim=-1) @ vh).transpose(1, 2).reshape(B, T, HD)


def prednet_mask(B, T, g):
    """causal | key padding, as rnnt_conv_transformer_lm.py:65-69 builds it; one batch element has every key but the first
    padded"""
    pad = torch.rand(B, T, generator=g) < 0.2
    pad[:, 0] = False
    pad[-1, 1:] = True
    return pad.unsqueeze(1).expand(B, T, T) | torch.triu(torch.ones(T, T, dtype=torch.bool), 1)


@pytest.mark.parametrize("B,T,H,D,masked", [(2, 200, 4, 64, False), (2, 100, 2, 128, False), (3, 51, 8, 64, True),
                                            (1, 130, 2, 64, True)])
def test_two_term_attention_forward(hip_device, B, T, H, D, masked):
    from pika_amd import gemm as G
    from pika_amd.model.hipops import PackedAttentionFn
    g = torch.Generator().manual_seed(B * 100 + T)
    qkv = torch.randn(B, T, 3 * H * D, generator=g)
    qkv[..., :H * D] *= 2.0
    mask = prednet_mask(B, T, g) if masked else None
    hi, lo = pair_on(hip_device, qkv.view(-1, 3 * H * D), 3 * H * D)
    hi, lo = hi.view(B, T, -1), lo.view(B, T, -1)
    # planes of ONE buffer, as the projection epilogue writes them
    buf = torch.empty((2,) + tuple(hi.shape), dtype=torch.bfloat16, device=hip_device)
    buf[0].copy_(hi)
    buf[1].copy_(lo)
    a = buf[0]
    a.requires_grad_(True)
    oh, ol = PackedAttentionFn.apply(a, H, 0.0, 0, buf[1], None if mask is None else mask.to(hip_device))
    x = two_term(qkv)
    HD = H * D
    want = attention_reference(x[..., :HD], x[..., HD:2 * HD], x[..., 2 * HD:], H, mask)
    got = oh.double().cpu() + ol.double().cpu()
    err = (got - want).abs().max().item()
    assert err < 1e-4 * want.abs().max().item() + 1e-5, err
    # one bf16 term is two orders worse
    o1 = PackedAttentionFn.apply(buf[0].detach(), H, 0.0, 0, None, None if mask is None else mask.to(hip_device))
    assert err < 0.05 * (o1.double().cpu() - want).abs().max().item()
    # backward through the hi planes (bf16 kernels), masked scores included
    w = torch.randn(B, T, HD, generator=g)
    (oh.float() * w.to(hip_device)).sum().backward()
    xr = qkv.double().requires_grad_(True)
    ref = attention_reference(xr[..., :HD], xr[..., HD:2 * HD], xr[..., 2 * HD:], H, mask)
    (ref * w.double()).sum().backward()
    rel = (a.grad.double().cpu() - xr.grad).norm() / xr.grad.norm()
    assert rel < 2e-2, rel


def test_masked_attention_bf16_kernels(hip_device):
    """The mask in the one-term kernels (forward, dQ, dK/dV): the prediction network's attention in the bf16 mode."""
    from pika_amd.model.hipops import AttentionFn
    B, T, H, D = 4, 51, 8, 64
    g = torch.Generator().manual_seed(9)
    q, k, v, w = [torch.randn(B, T, H * D, generator=g) for _ in range(4)]
    mask = prednet_mask(B, T, g)
    dev = [t.to(hip_device).requires_grad_(True) for t in (q, k, v)]
    out = AttentionFn.apply(*dev, H, 0.0, 0, mask.to(hip_device))
    (out * w.to(hip_device)).sum().backward()
    ref_in = [t.double().requires_grad_(True) for t in (q, k, v)]
    ref = attention_reference(*ref_in, H, mask)
    (ref * w.double()).sum().backward()
    assert (out.double().cpu() - ref.detach()).abs().max() < 3e-2 * ref.detach().abs().max() + 1e-3
    for a, b in zip(dev, ref_in):
        assert (a.grad.double().cpu() - b.grad).norm() / b.grad.norm() < 1.5e-2
    # a fully padded row (every key but the first masked) attends to key 0 only
    assert torch.isfinite(out).all()


def test_mixed_mode_layers_match_fp32_mode(hip_device):
    """One transformer layer + one TDNN/BatchNorm layer of the encoder width in "mixed" against the exact mode: forward to
    1e-4, parameter gradients to a few 1e-2 (bf16 backward)."""
    from pika_amd import gemm as G
    from pika_amd.model import ops
    from pika_amd.model.modules import TransformerEncoderLayer
    torch.manual_seed(1)
    layer = TransformerEncoderLayer(256, 4, 1024, 0.0).to(hip_device).train()
    conv = torch.nn.Conv2d(1, 256, (3, 256), dilation=(3, 1)).to(hip_device)
    bn = torch.nn.BatchNorm1d(256).to(hip_device).train()
    x = torch.randn(40, 300, 256, device=hip_device)

    def run(mode):
        old, G.PRECISION = G.PRECISION, mode
        try:
            for m in (layer, conv, bn):
                m.zero_grad(set_to_none=True)
            bn.running_mean.zero_()
            bn.running_var.fill_(1.0)
            h = x.clone().requires_grad_(True)
            if ops.tdnn_bn_ok(h, conv.weight, bn):
                y = ops.tdnn_bn(h, conv, bn, mfma_only=False)
            else:
                y = ops.batch_norm(ops.tdnn(h, conv.weight, conv.bias, 3, 1, relu=2).reshape(-1, 256), bn,
                                   relu_input=True).view(40, -1, 256)
            out = layer(y, mask=None)
            out.square().mean().backward()
            return out.detach().double(), {n: p.grad.double() for n, p in list(layer.named_parameters()) +
                                           [("conv.w", conv.weight), ("bn.w", bn.weight)]}, h.grad.double()
        finally:
            G.PRECISION = old
    o32, g32, x32 = run("fp32")
    om, gm, xm = run("mixed")
    assert ((om - o32).abs().max() / o32.abs().max()).item() < 1e-4
    gmax = max(g.norm().item() for g in g32.values())
    for n in g32:
        if g32[n].norm().item() < 1e-5 * gmax:      # zero up to rounding (key biases: the softmax is shift-invariant)
            continue
        rel = ((gm[n] - g32[n]).norm() / g32[n].norm()).item()
        assert rel < 3e-2, (n, rel)
    assert ((xm - x32).norm() / x32.norm()).item() < 3e-2
    o16, _, _ = run("bf16")
    assert ((o16 - o32).abs().max() / o32.abs().max()).item() > 10 * ((om - o32).abs().max() / o32.abs().max()).item()


def test_mixed_linear_with_an_output_width_the_two_term_kernel_does_not_take(hip_device):
    """N % 4 != 0 (the LAS output projection: V + 2 = 5002 classes): the product falls back to the exact kernel instead of
    failing, forward and backward."""
    from pika_amd import gemm as G
    from pika_amd.model import ops
    old, G.PRECISION = G.PRECISION, "mixed"
    try:
        torch.manual_seed(3)
        x = torch.randn(70, 128, device=hip_device, requires_grad=True)
        w = (torch.randn(5002, 128, device=hip_device) * 0.1).requires_grad_()
        b = torch.randn(5002, device=hip_device, requires_grad=True)
        y = ops.linear(x, w, b)
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        assert (y.double() - ref).abs().max().item() < 1e-4 * ref.abs().max().item()
        y.sum().backward()
        assert x.grad is not None and w.grad.shape == w.shape and b.grad.shape == b.shape
    finally:
        G.PRECISION = old


@pytest.mark.parametrize("M,N,K,taps", [(8192, 2048, 1024, 1), (16384, 1024, 3072, 1), (8000, 1536, 768, 3)])
def test_fp16_two_term_products(hip_device, M, N, K, taps):
    """precision "fp16x2": two fp16 terms per operand as three K-concatenated segments on the direct-to-LDS kernel
    (v_mfma_f32_16x16x32_f16): an fp32 product to ~2^-22 -- against float64, beside the exact six-segment mode; operands with
    a wide dynamic range (small elements whose low terms would be fp16 subnormals without the scaling) and a time-delay view."""
    from pika_amd import gemm as G
    g = torch.Generator().manual_seed(M + N + K)
    if taps == 1:
        a = torch.randn(M, K, generator=g) * torch.logspace(-3, 1.5, K).unsqueeze(0)
        a_op_src = a.to(hip_device)
        a_mat = a.double()
    else:
        C = K // taps
        x = (torch.randn(1, M + taps - 1, C, generator=g) * 3).to(hip_device)
        a_op_src = x
        a_mat = torch.cat([x[0, t:t + M].double().cpu() for t in range(taps)], 1)
    w = torch.randn(N, K, generator=g) * 0.03 * torch.logspace(-2, 0.5, N).unsqueeze(1)
    bias = torch.randn(N, generator=g)
    want = a_mat @ w.double().t() + bias.double()
    scale = want.abs().max().item()
    got = {}
    for prec in ("fp16x2", "fp32"):
        before = dict(G.FP16X2_STATS)
        if taps == 1:
            out = G.gemm_nt(a_op_src, w.to(hip_device), bias=bias.to(hip_device), precision=prec)
        else:
            a_op, rows, Kk, _ = G.time_delay(a_op_src, taps)
            w_d, bias_d = w.to(hip_device), bias.to(hip_device)      # descriptors hold raw pointers: keep the tensors
            b_op, _, _ = G.matrix(w_d)
            out = torch.empty(M, N, device=hip_device)
            G.launch(a_op, b_op, out, N, M, N, K, bias=bias_d, precision=prec)
        if prec == "fp16x2":
            assert G.FP16X2_STATS["fast"] == before["fast"] + 1            # the three-segment fp16 path took it
        got[prec] = (out.double().cpu() - want).abs().max().item() / scale
    # both sit at the fp32 accumulation noise of a K-term sum (measured: 1.5e-6 vs 2.5e-6 at K = 3072)
    assert got["fp16x2"] < 5e-6 and got["fp32"] < 5e-6 and got["fp16x2"] < 2 * got["fp32"] + 5e-7, got
    # an operand beyond fp16's range saturates instead of producing inf / nan
    a = torch.randn(4096, 1024, generator=g)
    a[3, 5] = 1e6
    before = G.FP16X2_STATS["fast"]
    out = G.gemm_nt(a.to(hip_device), torch.randn(2560, 1024, generator=g).to(hip_device) * 0.03, precision="fp16x2")
    assert G.FP16X2_STATS["fast"] == before + 1 and bool(torch.isfinite(out).all())
