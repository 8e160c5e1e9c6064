"""inf-norm clipping + Nesterov SGD of the training scripts
(train_transducer_bmuf_otfaug.py:105-110) as three multi-tensor HIP launches (pika_amd/optim.py) against the stock
torch calls on the same tensors: identical parameters, momentum buffers, gradients and returned norm over several
steps, for whole tensors and for views of one flat vector at odd offsets (how parameters live under BMUF)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(dev, flat_views):
    g = torch.Generator().manual_seed(12)
    shapes = [(1024, 240), (1024,), (7,), (3, 5, 11), (4096, 1024), (1,), (513,)]
    if flat_views:
        n = sum(int(torch.tensor(s).prod()) for s in shapes) + 3
        flat = torch.randn(n, generator=g).to(dev)
        out, off = [], 3                          # start 12 bytes off 16-byte alignment
        for s in shapes:
            k = int(torch.tensor(s).prod())
            out.append(torch.nn.Parameter(flat[off:off + k].view(s)))
            off += k
        return out
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]


@pytest.mark.parametrize("flat_views", [False, True])
@pytest.mark.parametrize("grad_scale", [0.01, 40.0])          # below / above the clip threshold of 3.0
def test_fused_clip_and_nesterov_sgd_equal_torch(hip_device, flat_views, grad_scale):
    from pika_amd import optim as O
    ref = _params(hip_device, flat_views)
    ours = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref = O._TorchSGD(ref, 0.003, momentum=0.9, nesterov=True)
    o_our = O.SGD(ours, 0.003, momentum=0.9, nesterov=True)
    g = torch.Generator().manual_seed(3)
    for step in range(4):
        for a, b in zip(ref, ours):
            gr = (torch.randn(a.shape, generator=g) * grad_scale).to(hip_device)
            a.grad, b.grad = gr.clone(), gr.clone()
        n_ref = O._torch_clip(ref, 3.0, norm_type=float("inf"))
        n_our = O.clip_grad_norm_(ours, 3.0, norm_type=float("inf"))
        assert torch.equal(n_ref, n_our)
        for a, b in zip(ref, ours):
            assert torch.allclose(a.grad, b.grad, rtol=2e-7, atol=0), step      # x * clamp(c, max=1) vs x * c
        o_ref.step()
        o_our.step()
        for a, b in zip(ref, ours):
            assert torch.allclose(a, b, rtol=0, atol=1e-7 * float(a.abs().max())), step
            mb = o_ref.state[a]["momentum_buffer"]
            assert torch.allclose(mb, o_our.state[b]["momentum_buffer"], rtol=0, atol=3e-7 * float(mb.abs().max()))
    # a rebuilt optimizer starts without momentum again (the scripts rebuild it after every BMUF sync, :121-123)
    o_our2 = O.SGD(ours, 0.003, momentum=0.9, nesterov=True)
    o_ref2 = O._TorchSGD(ref, 0.003, momentum=0.9, nesterov=True)
    for a, b in zip(ref, ours):          # fresh gradients: torch's multi-tensor Nesterov step leaves g + momentum*buf in .grad
        gr = (torch.randn(a.shape, generator=g) * grad_scale).to(hip_device)
        a.grad, b.grad = gr.clone(), gr.clone()
    o_our2.step()
    o_ref2.step()
    for a, b in zip(ref, ours):
        assert torch.allclose(a, b, rtol=0, atol=2e-7 * float(a.abs().max()))


def test_install_routes_the_script_calls_and_leaves_the_rest_to_torch(hip_device):
    from pika_amd import optim as O
    O.install()
    try:
        p = [torch.nn.Parameter(torch.randn(300, 7, device=hip_device))]
        p[0].grad = torch.randn(300, 7, device=hip_device) * 10
        ref = float(p[0].grad.abs().max())
        n = torch.nn.utils.clip_grad_norm_(p, 3.0, norm_type=float("inf"))
        assert abs(float(n) - ref) < 1e-6 and abs(float(p[0].grad.abs().max()) - 3.0) < 1e-4
        assert isinstance(torch.optim.SGD(p, 0.1, momentum=0.9, nesterov=True), O._TorchSGD)
        # other norm types / CPU tensors / plain SGD: stock implementations
        n2 = torch.nn.utils.clip_grad_norm_(p, 1.0)                      # 2-norm
        assert abs(float(p[0].grad.norm()) - 1.0) < 1e-4 and float(n2) > 1.0
        q = [torch.nn.Parameter(torch.randn(5))]
        q[0].grad = torch.ones(5)
        before = q[0].detach().clone()
        torch.optim.SGD(q, 0.5).step()
        assert torch.allclose(q[0], before - 0.5)
        nan = [torch.nn.Parameter(torch.zeros(4, device=hip_device))]
        nan[0].grad = torch.tensor([1.0, float("nan"), 2.0, 3.0], device=hip_device)
        assert torch.isnan(torch.nn.utils.clip_grad_norm_(nan, 3.0, norm_type=float("inf")))
    finally:
        O.uninstall()
