"""The persistent LSTM recurrence of training (include/pika_lstm.h, pika_amd/model/lstm.py) against torch's nn.LSTM in fp32
on the same device: outputs, input gradient and every parameter gradient, at the prediction network's own shape
(reference trainer/model/transducer.py:55-61,93-96: B x (U + 1) labels, H = 1024, two layers) and at ragged shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets(E, H, layers, seed, dropout=0.0):
    torch.manual_seed(seed)
    ref = torch.nn.LSTM(E, H, layers, batch_first=True, dropout=dropout).cuda()
    return ref


def _run(ref, x, dy, persistent, mode):
    from pika_amd import gemm as G
    from pika_amd.model import lstm, transducer
    old, lstm.PERSISTENT, oldp = lstm.PERSISTENT, persistent, G.PRECISION
    G.PRECISION = mode
    try:
        ref.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        took = lstm.applies(ref, xi)
        out = transducer._lstm_forward(ref, xi)
        out.backward(dy)
        torch.cuda.synchronize()
        return took, out.detach(), xi.grad.detach(), {n: p.grad.detach().clone() for n, p in ref.named_parameters()}
    finally:
        lstm.PERSISTENT, G.PRECISION = old, oldp


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("B,S,E,H,layers", [(32, 51, 100, 1024, 2), (5, 7, 36, 256, 1), (19, 3, 64, 512, 3), (48, 1, 100, 768, 2)])
def test_recurrence_matches_torch_lstm(B, S, E, H, layers):
    from pika_amd.model import lstm
    ref = _nets(E, H, layers, seed=B + S)
    ref.train()
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(B, S, E, device="cuda", generator=g)
    dy = torch.randn(B, S, H, device="cuda", generator=g)
    took, out, dx, grads = _run(ref, x, dy, True, "bf16x3")
    assert took, "the persistent recurrence did not take the call"
    assert lstm.status() == 0
    with torch.no_grad():
        want, _ = ref(x)
    xi = x.clone().requires_grad_(True)
    ref.zero_grad(set_to_none=True)
    o2, _ = ref(xi)
    o2.backward(dy)
    assert _rel(out, want) < 2e-4, _rel(out, want)
    assert _rel(dx, xi.grad) < 1e-3, _rel(dx, xi.grad)
    for n, p in ref.named_parameters():
        assert _rel(grads[n], p.grad) < 1e-3, (n, _rel(grads[n], p.grad))


def test_fp32_mode_keeps_the_library_recurrence():
    from pika_amd.model import lstm
    ref = _nets(100, 1024, 2, seed=3)
    x = torch.randn(4, 5, 100, device="cuda")
    took, out, _, _ = _run(ref, x, torch.ones(4, 5, 1024, device="cuda"), True, "fp32")
    assert not took
    with torch.no_grad():
        assert torch.equal(out, ref(x)[0])
    assert not lstm.applies(ref, x.double())


def test_oversized_batch_falls_back():
    from pika_amd import gemm as G
    from pika_amd.model import lstm
    ref = _nets(100, 1024, 1, seed=4)
    old = G.PRECISION
    G.PRECISION = "mixed"
    try:
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        rows = 16 * (cus // 64) + 1                    # one row block more than fits
        assert not lstm.applies(ref, torch.zeros(rows, 2, 100, device="cuda"))
        assert lstm.applies(ref, torch.zeros(rows - 1, 2, 100, device="cuda"))
    finally:
        G.PRECISION = old


def test_c_abi_refuses_bad_arguments():
    from pika_amd import _lib
    lib = _lib.lib()
    assert lib.pika_lstm_train_packed_bytes(1000) == -1 and lib.pika_lstm_train_packed_bytes(2048) == -1
    assert lib.pika_lstm_train_packed_bytes(1024) == 2 * 4 * 1024 * 1024 * 4
    assert lib.pika_lstm_train_fwd_work_bytes(0, 4, 256) == -1 and lib.pika_lstm_train_bwd_work_bytes(4, 0, 256) == -1
    assert lib.pika_lstm_train_bwd_work_bytes(51, 32, 1024) == 256 + 4 * 51 * 2 * 64 * 64 * 256
    assert lib.pika_lstm_train_fwd(None, None, None, None, None, None, 0, 4, 4, 256, None) == -1
    assert lib.pika_lstm_train_bwd(None, None, None, None, None, None, 0, 1, 4, 4, 256, None) == -1


def test_forward_hooks_of_the_module_see_the_call():
    from pika_amd import gemm as G
    from pika_amd.model import transducer
    ref = _nets(100, 1024, 2, seed=5)
    seen = {}
    h = ref.register_forward_hook(lambda m, i, o: seen.update(out=o[0].detach(), h_n=o[1][0], c_n=o[1][1], x=i[0]))
    old = G.PRECISION
    G.PRECISION = "bf16x3"
    try:
        x = torch.randn(3, 6, 100, device="cuda")
        out = transducer._lstm_forward(ref, x)
    finally:
        G.PRECISION = old
        h.remove()
    with torch.no_grad():
        want, (h_n, c_n) = ref(x)
    assert seen["x"] is x and torch.equal(seen["out"], out.detach())
    assert _rel(seen["out"], want) < 2e-4 and _rel(seen["h_n"], h_n) < 2e-4 and _rel(seen["c_n"], c_n) < 2e-4


def test_graphed_backward_gives_summed_parameters_their_own_gradient_buffers():
    """b_ih + b_hh enters the model as a sum: autograd.grad hands both parameters ONE tensor; the captured backward must
    not (the clip scales gradients in place)."""
    from pika_amd import train_graph
    a = torch.ones(4, device="cuda")
    g = [a, None, a, a.clone()]
    out = train_graph.distinct_buffers(g)
    ptrs = [t.data_ptr() for t in out if t is not None]
    assert out[1] is None and len(set(ptrs)) == 3 and all(torch.equal(t, a) for t in out if t is not None)


def test_backward_scratch_is_left_as_a_memset_leaves_it():
    """`armed`: a completed backward launch resets every exchange word it read, over changing shapes too."""
    from pika_amd.model import lstm
    for B, S, H in ((32, 51, 1024), (7, 9, 256), (20, 4, 512)):
        ref = _nets(100, H, 2, seed=S)
        x = torch.randn(B, S, 100, device="cuda")
        _run(ref, x, torch.randn(B, S, H, device="cuda"), True, "bf16x3")
        w = lstm._WORK[(0, True)]
        assert lstm.status() == 0
        assert bool((w[256:] == 255).all()), (B, S, H)


def test_no_grad_forward_and_growing_shapes_keep_earlier_scratch_alive():
    """Evaluation (no autograd) takes the same launch; a larger shape later gets a larger scratch while the outgrown one
    stays allocated (a captured training step may hold its address)."""
    from pika_amd import gemm as G
    from pika_amd.model import lstm, transducer
    ref = _nets(100, 512, 2, seed=9).eval()
    old = G.PRECISION
    G.PRECISION = "mixed"
    try:
        with torch.no_grad():
            x = torch.randn(4, 6, 100, device="cuda")
            assert lstm.applies(ref, x)
            got = transducer._lstm_forward(ref, x)
            assert _rel(got, ref(x)[0]) < 2e-4
            first = lstm._WORK[(0, True)]
            big = torch.randn(33, 40, 100, device="cuda")
            got = transducer._lstm_forward(ref, big)
            assert _rel(got, ref(big)[0]) < 2e-4
        now = lstm._WORK[(0, True)]
        if now is not first:
            assert any(w is first for w in lstm._RETIRED) and now.numel() >= first.numel() + first.numel() // 2
        assert lstm.status() == 0
    finally:
        G.PRECISION = old
