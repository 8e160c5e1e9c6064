"""Fused self-attention (include/pika_attn.h) vs the fp64 formula of
trainer/model/multi_headed_attn.py:199-231 (query / sqrt(D), softmax, dropout on the probabilities,
context = drop_attn @ value)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def reference(q, k, v, heads, keep=None, inv_keep=1.0):
    B, T, HD = q.shape
    D = HD // heads
    qh = (q / math.sqrt(D)).view(B, T, heads, D).transpose(1, 2)
    kh = k.view(B, T, heads, D).transpose(1, 2)
    vh = v.view(B, T, heads, D).transpose(1, 2)
    attn = torch.softmax(qh @ kh.transpose(2, 3), dim=-1)
    if keep is not None:
        attn = attn * keep.view(B, heads, T, T).to(attn.dtype) * inv_keep
    return (attn @ vh).transpose(1, 2).reshape(B, T, HD)


def bf16r(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("B,T,H,D", [(1, 1, 1, 64), (2, 64, 2, 64), (2, 100, 3, 64), (1, 77, 2, 128),
                                     (2, 333, 4, 128), (3, 250, 16, 64)])
@pytest.mark.parametrize("p_drop", [0.0, 0.2])
def test_attention_forward_backward(hip_device, B, T, H, D, p_drop):
    from pika_amd.model.hipops import AttentionFn, attention_keep_mask
    g = torch.Generator().manual_seed(B * 1000 + T + H)
    q, k, v, w = [torch.randn(B, T, H * D, generator=g) for _ in range(4)]
    q = q * 2.0   # sharper softmax
    seed = 1234 + T
    dev = [t.to(hip_device).requires_grad_(True) for t in (q, k, v)]
    out = AttentionFn.apply(*dev, H, p_drop, seed)
    (out * w.to(hip_device)).sum().backward()
    keep, inv_keep = None, 1.0
    if p_drop > 0:
        keep = attention_keep_mask(B * H, T, p_drop, seed, hip_device).cpu()
        thr = round(p_drop * 65536)
        inv_keep = 65536.0 / (65536 - thr)
        if B * H * T * T > 20000:
            assert abs(keep.float().mean().item() - (1 - p_drop)) < 4 * math.sqrt(0.16 / keep.numel()) + 1e-4
        # rows and columns are decorrelated
        if T >= 64:
            assert abs(keep.float().mean(dim=-1).std().item() - math.sqrt(0.16 / T)) < 0.3 * math.sqrt(0.16 / T)
    # the kernel rounds q/sqrt(D)*log2e, k, v (and P, dS, dO) to bf16: compare with the fp64 formula on
    # the same inputs, tolerance = bf16 operand rounding through a softmax-weighted average
    ref_in = [t.double().requires_grad_(True) for t in (q, k, v)]
    ref = reference(*ref_in, H, keep, inv_keep)
    (ref * w.double()).sum().backward()
    err = (out.double().cpu() - ref.detach()).abs().max().item()
    assert err < 3e-2 * ref.detach().abs().max().item() + 1e-3, err
    for a, b in zip(dev, ref_in):
        s = b.grad.abs().max().item()
        e = (a.grad.double().cpu() - b.grad).abs().max().item()
        assert e < 4e-2 * max(s, 0.5), (e, s)   # T=1: exact gradient is 0, error = bf16 rounding of dO.v
        # rounding is unbiased: relative L2 error well below the max-norm bound
        if s > 0:
            rel = (a.grad.double().cpu() - b.grad).norm() / b.grad.norm()
            assert rel < 1.5e-2, rel


def test_attention_module_path(hip_device):
    """ops.attention routes the encoder's case to the fused kernels (bf16 mode), the parity mode and
    masked calls to the reference chain; eval mode has no dropout."""
    from pika_amd import gemm as G
    from pika_amd.model import ops
    g = torch.Generator().manual_seed(5)
    q, k, v = [torch.randn(2, 90, 256, generator=g).to(hip_device) for _ in range(3)]
    old = G.PRECISION
    try:
        G.PRECISION = "fp32"
        want = ops.attention(q, k, v, 4, None, 0.2, False)
        G.PRECISION = "bf16"
        got = ops.attention(q, k, v, 4, None, 0.2, False)
        assert (got - want).abs().max() < 3e-2
        torch.manual_seed(3)
        a = ops.attention(q, k, v, 4, None, 0.2, True)
        torch.manual_seed(3)
        b = ops.attention(q, k, v, 4, None, 0.2, True)
        assert torch.equal(a, b) and not torch.equal(a, got)
    finally:
        G.PRECISION = old


def test_attention_argument_errors(hip_device):
    from pika_amd.model.hipops import AttentionFn
    x = torch.randn(1, 8, 96, device=hip_device)
    with pytest.raises(RuntimeError):
        AttentionFn.apply(x, x, x, 2, 0.0, 0)      # D = 48
    y = torch.randn(1, 8, 128, device=hip_device)
    with pytest.raises(RuntimeError):
        AttentionFn.apply(y, y, y, 2, 1.0, 0)      # p_drop out of range


def test_packed_attention_matches_separate(hip_device):
    """PackedAttentionFn on [q | k | v] (pitch 3*H*D) is bit-identical to AttentionFn on the three blocks,
    forward and backward, dropout included."""
    from pika_amd.model.hipops import AttentionFn, PackedAttentionFn
    g = torch.Generator().manual_seed(11)
    B, T, H, D = 2, 150, 4, 64
    qkv = torch.randn(B, T, 3 * H * D, generator=g).to(hip_device)
    w = torch.randn(B, T, H * D, generator=g).to(hip_device)
    a = qkv.clone().requires_grad_(True)
    out_p = PackedAttentionFn.apply(a, H, 0.2, 99)
    (out_p * w).sum().backward()
    parts = [qkv[..., i * H * D:(i + 1) * H * D].clone().requires_grad_(True) for i in range(3)]
    out_s = AttentionFn.apply(*parts, H, 0.2, 99)
    (out_s * w).sum().backward()
    assert torch.equal(out_p, out_s)
    assert torch.equal(a.grad, torch.cat([p.grad for p in parts], -1))


@pytest.mark.parametrize("B,T,H,D,masked", [(2, 300, 4, 64, False), (1, 130, 2, 128, False), (2, 77, 2, 64, True)])
def test_inference_attention_on_two_fp16_terms(hip_device, B, T, H, D, masked):
    """pika_attention_infer_f16x2 (the decoder's encoder pass): softmax(q k^T / sqrt(D)) v with q, k, v as two fp16 terms
    (22 mantissa bits) against float64 -- fp32-grade: an order of magnitude inside what two bf16 terms give."""
    from pika_amd import gemm as G
    from pika_amd.model import hipops
    g = torch.Generator().manual_seed(B * 1000 + T)
    q, k, v = (torch.randn(B, T, H * D, generator=g) * s for s in (2.0, 2.0, 1.0))
    mask = None
    if masked:
        mask = torch.triu(torch.ones(T, T, dtype=torch.bool), 1).unsqueeze(0).expand(B, T, T).clone()
        mask[:, :, T - 9:] = True
        mask[:, :, 0] = False
    qh = (q.double() / D ** 0.5).view(B, T, H, D).transpose(1, 2)
    sc = qh @ k.double().view(B, T, H, D).transpose(1, 2).transpose(2, 3)
    if mask is not None:
        sc = sc.masked_fill(mask.unsqueeze(1), -1e18)
    want = (torch.softmax(sc, -1) @ v.double().view(B, T, H, D).transpose(1, 2)).transpose(1, 2).reshape(B, T, H * D)
    out = {}
    old = G.PRECISION
    try:
        for mode in ("fp16x2", "bf16x3"):
            G.PRECISION = mode
            with torch.no_grad():
                assert hipops.attention_infer_ok(q.to(hip_device), k.to(hip_device), v.to(hip_device), H,
                                                 None if mask is None else mask.to(hip_device))
                got = hipops.attention_infer_two_term(q.to(hip_device), k.to(hip_device), v.to(hip_device), H,
                                                      None if mask is None else mask.to(hip_device))
                # the same on the packed [q | k | v] projection, planes from one split launch: the very same bits
                packed = hipops.attention_infer_packed(torch.cat((q, k, v), -1).to(hip_device), H,
                                                       None if mask is None else mask.to(hip_device))
                assert torch.equal(got, packed), mode
            out[mode] = float((got.double().cpu() - want).abs().max() / want.abs().max())
    finally:
        G.PRECISION = old
    print("inference attention vs float64: two fp16 terms %.1e, two bf16 terms %.1e" % (out["fp16x2"], out["bf16x3"]))
    assert out["fp16x2"] < 2e-6 and out["bf16x3"] < 1e-4, out
