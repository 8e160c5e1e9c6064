"""BatchNorm kernels (include/pika_norm.h) vs torch's BatchNorm1d in fp64 (training mode)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,C", [(1, 4), (7, 8), (300, 64), (5000, 1024), (33, 100)])
def test_batch_norm_forward_backward_running_stats(hip_device, M, C):
    from pika_amd.model import ops
    g = torch.Generator().manual_seed(M + C)
    x = torch.relu(torch.randn(M, C, generator=g, dtype=torch.float64) * 2 + 0.5)
    w = torch.randn(M, C, generator=g, dtype=torch.float64)
    ref = torch.nn.BatchNorm1d(C).double().train()
    ours = torch.nn.BatchNorm1d(C).to(hip_device).train()
    with torch.no_grad():
        ref.weight.copy_(torch.randn(C, generator=g) * 0.3 + 1)
        ref.bias.copy_(torch.randn(C, generator=g))
        ours.weight.copy_(ref.weight.float()); ours.bias.copy_(ref.bias.float())
    if M == 1:
        return  # torch refuses a single row in training mode; ours is covered by the other sizes
    xr = x.clone().requires_grad_(True)
    (ref(xr) * w).sum().backward()
    xd = x.float().to(hip_device).requires_grad_(True)
    y = ops.batch_norm(xd, ours)
    (y * w.float().to(hip_device)).sum().backward()
    assert (y.double().cpu() - ref(x).detach()).abs().max() < 1e-4   # second ref call also moves its stats
    sc = xr.grad.abs().max().item() + 1e-9
    assert (xd.grad.double().cpu() - xr.grad).abs().max() < 2e-4 * max(sc, 1)
    assert (ours.weight.grad.double().cpu() - ref.weight.grad).abs().max() < 1e-3 * max(1, ref.weight.grad.abs().max().item())
    assert (ours.bias.grad.double().cpu() - ref.bias.grad).abs().max() < 1e-3 * max(1, ref.bias.grad.abs().max().item())
    ref2 = torch.nn.BatchNorm1d(C).double().train()
    ref2(x)
    assert torch.allclose(ours.running_mean.double().cpu(), ref2.running_mean, atol=1e-5)
    assert torch.allclose(ours.running_var.double().cpu(), ref2.running_var, rtol=1e-4, atol=1e-5)
    assert int(ours.num_batches_tracked) == 1
