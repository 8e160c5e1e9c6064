"""Joint-network kernels (include/pika_joint.h) vs torch fp64 formulas of
trainer/model/transducer.py:98-111."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,T,U,H", [(1, 1, 1, 4), (2, 5, 3, 64), (3, 17, 7, 1024), (2, 9, 4, 100)])
def test_gate_forward_backward(hip_device, B, T, U, H):
    from pika_amd import gemm as G
    from pika_amd.model.hipops import GateFn
    old, G.PRECISION = G.PRECISION, "fp32"
    try:
        g = torch.Generator().manual_seed(B * 100 + T)
        ins = [torch.randn(B, n, H, generator=g, dtype=torch.float64) for n in (T, U, T, U)]
        w = torch.randn(B, T, U, H, generator=g, dtype=torch.float64)
        ref_in = [t.clone().requires_grad_(True) for t in ins]
        e1, p1, eg, pg = ref_in
        ref = torch.tanh(e1.unsqueeze(2) + p1.unsqueeze(1)) * torch.sigmoid(eg.unsqueeze(2) + pg.unsqueeze(1))
        (ref * w).sum().backward()
        dev_in = [t.float().to(hip_device).requires_grad_(True) for t in ins]
        h = GateFn.apply(*dev_in)
        assert h.dtype == torch.float32
        assert (h.double().cpu() - ref.detach()).abs().max() < 2e-6
        (h * w.float().to(hip_device)).sum().backward()
        for a, b in zip(dev_in, ref_in):
            scale = b.grad.abs().max().item()
            assert (a.grad.double().cpu() - b.grad).abs().max() < 1e-5 * max(scale, 1.0)
        G.PRECISION = "bf16"
        hb = GateFn.apply(*[t.detach() for t in dev_in])
        assert hb.dtype == torch.bfloat16
        assert (hb.double().cpu() - ref.detach()).abs().max() < 2 ** -8
    finally:
        G.PRECISION = old


@pytest.mark.parametrize("rows,cols,scale", [(1, 4, 1.0), (7, 100, 1.0), (33, 5000, 0.8), (5, 13, 1.0),
                                             (3, 9000, 1.0), (4, 8192, 0.5)])
def test_log_softmax_rows_forward_backward(hip_device, rows, cols, scale):
    from pika_amd.model.hipops import LogSoftmaxFn
    g = torch.Generator().manual_seed(rows * 31 + cols)
    x = torch.randn(rows, cols, generator=g, dtype=torch.float64) * 3
    w = torch.randn(rows, cols, generator=g, dtype=torch.float64)
    w[w.abs() < 1.0] = 0.0  # sparse upstream gradient, like the RNN-T loss
    xr = x.clone().requires_grad_(True)
    ref = torch.log_softmax(scale * xr, dim=-1)
    (ref * w).sum().backward()
    xd = x.float().to(hip_device).requires_grad_(True)
    out = LogSoftmaxFn.apply(xd * 1.0, scale)  # *1.0: a non-leaf buffer the op may overwrite
    assert (out.double().cpu() - ref.detach()).abs().max() < 1e-5
    (out * w.float().to(hip_device)).sum().backward()
    assert (xd.grad.double().cpu() - xr.grad).abs().max() < 1e-5 * max(1.0, w.abs().sum(-1).max().item())
