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


@pytest.mark.parametrize("rows,C", [(1, 4), (7, 100), (300, 1024), (5000, 1024), (33, 2048), (65, 64)])
@pytest.mark.parametrize("bf16", [False, True])
def test_layer_norm_forward_backward(hip_device, rows, C, bf16):
    """LayerNormFn (include/pika_norm.h) vs torch fp64 layer_norm; the bf16 variant rounds the output once
    and accepts a bf16 incoming gradient (what the MFMA consumers would do to an fp32 tensor anyway)."""
    import torch.nn.functional as F
    from pika_amd.model.hipops import LayerNormFn
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 2 + 0.5
    w = torch.randn(C, generator=g)
    b = torch.randn(C, generator=g)
    gy = torch.randn(rows, C, generator=g)
    if bf16:
        gy = gy.bfloat16().float()
    ref = [t.double().requires_grad_(True) for t in (x, w, b)]
    yr = F.layer_norm(ref[0], (C,), ref[1], ref[2], 1e-6)
    (yr * gy.double()).sum().backward()
    dev = [t.to(hip_device).requires_grad_(True) for t in (x, w, b)]
    y = LayerNormFn.apply(dev[0], dev[1], dev[2], 1e-6, bf16)
    assert y.dtype == (torch.bfloat16 if bf16 else torch.float32)
    tol = 2 ** -8 if bf16 else 2e-6
    assert (y.double().cpu() - yr.detach()).abs().max() < tol * max(1.0, yr.detach().abs().max().item())
    (y * gy.to(hip_device).to(y.dtype)).sum().backward()
    for a, r in zip(dev, ref):
        s = r.grad.abs().max().item()
        assert (a.grad.double().cpu() - r.grad).abs().max() < 2e-5 * max(s, 1.0) * (rows ** 0.5 if a.dim() == 1 else 1.0)


def test_layer_norm_backward_parameter_gradients_with_and_without_the_scratch(hip_device):
    """pika_layer_norm_bwd at the benchmark's shape (31808 rows x 512): with `partials` the workgroups' column sums are added up
    in a fixed order by a second launch -- two calls give the SAME bits -- and equal the float-atomics form (partials NULL)
    and the float64 sums to rounding."""
    from pika_amd import _lib
    from pika_amd import gemm as G
    rows, C = 31808, 512
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.5).to(hip_device)
    w = torch.randn(C, generator=g).to(hip_device)
    dy = torch.randn(rows, C, generator=g).bfloat16().to(hip_device)
    mean = x.mean(1).contiguous()
    rstd = (x.var(1, unbiased=False) + 1e-6).rsqrt().contiguous()
    lib = _lib.lib()
    n = int(lib.pika_layer_norm_bwd_partial_floats(rows, C))
    assert n >= 2 * C and n % (2 * C) == 0

    def run(with_scratch):
        dx, dg, db = torch.empty_like(x), torch.full((C,), 7.0, device=hip_device), torch.full((C,), 7.0, device=hip_device)
        part = torch.empty(n, device=hip_device) if with_scratch else None
        with torch.cuda.device(hip_device):
            _lib.check(lib.pika_layer_norm_bwd(dy.data_ptr(), G.PIKA_BF16, x.data_ptr(), rows, C, w.data_ptr(), mean.data_ptr(),
                                               rstd.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                               None if part is None else part.data_ptr(), None,
                                               torch.cuda.current_stream().cuda_stream), "pika_layer_norm_bwd")
        torch.cuda.synchronize()
        return dx, dg, db
    a, b, c = run(True), run(True), run(False)
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    assert torch.equal(a[0], c[0])
    xh = ((x - mean[:, None]) * rstd[:, None]).double()
    dg64, db64 = (dy.double() * xh).sum(0), dy.double().sum(0)
    for got in (a, c):
        assert (got[1].double() - dg64).abs().max() < 1e-5 * rows ** 0.5 * 4
        assert (got[2].double() - db64).abs().max() < 1e-5 * rows ** 0.5 * 4


@pytest.mark.parametrize("B,Tp,C,V,sub,div", [(3, 40, 64, 31, 0, 1), (2, 25, 128, 140, 39, 4), (4, 16, 8, 16, 0, 1)])
def test_batch_norm_over_the_data_rows_of_a_padded_time_axis(hip_device, B, Tp, C, V, sub, div):
    """pika_bn_valid_t: the matrix is B blocks of Tp rows of which the first (V - sub) // div hold data (V read on the
    device).  Statistics, running statistics, outputs and all three gradients equal those of BatchNorm1d (float64) over
    the data rows alone; padding rows come out as zeros and receive no gradient -- whatever they hold."""
    from pika_amd.model import ops
    vl = min(max((V - sub) // div, 0), Tp)
    g = torch.Generator().manual_seed(B * 100 + Tp)
    x = torch.relu(torch.randn(B, Tp, C, generator=g, dtype=torch.float64) * 2 + 0.5)
    x[:, vl:] = torch.randn(B, Tp - vl, C, generator=g, dtype=torch.float64) * 50        # garbage in the padding
    w = torch.randn(B, Tp, C, generator=g, dtype=torch.float64)
    w[:, vl:] = 0                               # nothing downstream sends gradient into padding rows
    ref = torch.nn.BatchNorm1d(C).double().train()
    ours = torch.nn.BatchNorm1d(C).to(hip_device).train()
    with torch.no_grad():
        ref.weight.copy_(torch.randn(C, generator=g) * 0.3 + 1)
        ref.bias.copy_(torch.randn(C, generator=g))
        ours.weight.copy_(ref.weight.float()); ours.bias.copy_(ref.bias.float())
    xr = x[:, :vl].reshape(-1, C).clone().requires_grad_(True)
    yr = ref(xr)
    (yr * w[:, :vl].reshape(-1, C)).sum().backward()
    t_valid = torch.tensor([V], dtype=torch.int32, device=hip_device)
    xd = x.float().to(hip_device).reshape(-1, C).requires_grad_(True)
    with ops.valid_rows(t_valid, Tp, sub, div):
        y = ops.batch_norm(xd, ours)
    (y * w.float().to(hip_device).reshape(-1, C)).sum().backward()
    y3 = y.double().cpu().view(B, Tp, C)
    assert (y3[:, :vl].reshape(-1, C) - yr.detach()).abs().max() < 1e-4
    assert bool((y3[:, vl:] == 0).all())
    gx = xd.grad.double().cpu().view(B, Tp, C)
    assert (gx[:, :vl].reshape(-1, C) - xr.grad).abs().max() < 2e-4 * max(1.0, xr.grad.abs().max().item())
    assert bool((gx[:, vl:] == 0).all())
    assert (ours.weight.grad.double().cpu() - ref.weight.grad).abs().max() < 1e-3 * max(1, ref.weight.grad.abs().max().item())
    assert (ours.bias.grad.double().cpu() - ref.bias.grad).abs().max() < 1e-3 * max(1, ref.bias.grad.abs().max().item())
    assert torch.allclose(ours.running_mean.double().cpu(), ref.running_mean, atol=1e-5)
    assert torch.allclose(ours.running_var.double().cpu(), ref.running_var, rtol=1e-4, atol=1e-5)
    # the SAME launches with another valid length on the device word: nothing of the length is a host value
    t_valid.fill_(sub + div * (vl - 1) if vl > 1 else V)
    vl2 = min(max((int(t_valid.item()) - sub) // div, 0), Tp)
    with ops.valid_rows(t_valid, Tp, sub, div):
        y2 = ops.batch_norm(xd.detach(), ours).double().cpu().view(B, Tp, C)
    ref3 = torch.nn.BatchNorm1d(C).double().train()
    with torch.no_grad():
        ref3.weight.copy_(ref.weight); ref3.bias.copy_(ref.bias)
    assert (y2[:, :vl2].reshape(-1, C) - ref3(x[:, :vl2].reshape(-1, C)).detach()).abs().max() < 1e-4


@pytest.mark.parametrize("rows,cols,bf16", [(31808, 512, True), (31808, 2048, True), (7680, 1536, False), (3, 8, True),
                                            (70001, 264, False), (257, 5001, True), (1000, 12, False)])
def test_column_sums_with_and_without_the_scratch(hip_device, rows, cols, bf16):
    """pika_colsum / pika_colsum_bf16 (bias gradients): chunk sums side by side + a fixed-order fold (the same bits on every
    call) against the float-atomics form (partials NULL; also what odd widths fall back to) and float64."""
    from pika_amd import _lib
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g)
    x = (x.bfloat16() if bf16 else x).to(hip_device)
    lib = _lib.lib()
    fn = lib.pika_colsum_bf16 if bf16 else lib.pika_colsum
    n = int(lib.pika_colsum_partial_floats(rows, cols))
    assert n >= cols and n % cols == 0 and n // cols <= 1024

    def run(with_scratch):
        out = torch.full((cols,), 3.0, device=hip_device)
        part = torch.empty(n, device=hip_device) if with_scratch else None
        with torch.cuda.device(hip_device):
            _lib.check(fn(x.data_ptr(), cols, rows, cols, out.data_ptr(), None if part is None else part.data_ptr(),
                          torch.cuda.current_stream().cuda_stream), "pika_colsum")
        torch.cuda.synchronize()
        return out
    a, b, c = run(True), run(True), run(False)
    ref = x.double().sum(0)
    tol = 2e-6 * rows ** 0.5 * max(1.0, float(x.float().abs().max()))
    assert (a.double() - ref).abs().max() < tol and (c.double() - ref).abs().max() < tol
    if cols % (8 if bf16 else 4) == 0:
        assert torch.equal(a, b)


@pytest.mark.parametrize("N,taps,C", [(512, 3, 1024), (1536, 1, 512), (64, 5, 100), (33, 2, 7), (2048, 1, 1024)])
def test_weight_operand_of_the_input_gradient_product_in_one_launch(hip_device, N, taps, C):
    """pika_weight_taps_transposed_bf16 == flip(taps) + per-tap transpose + bf16 cast as torch computes them."""
    from pika_amd.model.hipops import weight_taps_transposed
    w = torch.randn(N, taps * C, generator=torch.Generator().manual_seed(N + C)).to(hip_device)
    want = w.view(N, taps, C).flip(1).permute(2, 1, 0).reshape(C, taps * N).to(torch.bfloat16)
    with torch.cuda.device(hip_device):
        got = weight_taps_transposed(w, taps)
    assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.parametrize("bf16", [False, True])
def test_layer_norm_with_the_skip_connection_gradient_added_in_the_backward_kernel(hip_device, bf16):
    """ops.layer_norm(..., with_skip=True): (LN(x), x') with x' an alias of x whose gradient reaches the LN node -- the sum
    d(LN)/dx + d(skip) is formed inside pika_layer_norm_bwd (dx_add), not by an accumulation launch.  Against autograd over
    the plain formulation in float64, and with only one of the two outputs used."""
    from pika_amd.model import ops
    rows, C = 300, 512
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, rows // 2, C, generator=g)
    ln = torch.nn.LayerNorm(C, eps=1e-6)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(C, generator=g)); ln.bias.copy_(torch.randn(C, generator=g))
    gy, gs = torch.randn(2, rows // 2, C, generator=g), torch.randn(2, rows // 2, C, generator=g)
    if bf16:
        gy = gy.bfloat16().float()
    ref_ln = torch.nn.LayerNorm(C, eps=1e-6).double()
    ref_ln.load_state_dict(ln.state_dict())
    for use_y, use_s in ((True, True), (True, False), (False, True)):
        xr = x.double().requires_grad_(True)
        loss = 0
        if use_y:
            loss = loss + (ref_ln(xr) * gy.double()).sum()
        if use_s:
            loss = loss + (xr * gs.double()).sum()
        ref_ln.zero_grad()
        loss.backward()
        dev_ln = torch.nn.LayerNorm(C, eps=1e-6).to(hip_device)
        dev_ln.load_state_dict(ln.state_dict())
        xd = x.to(hip_device).requires_grad_(True)
        y, skip = ops.layer_norm(xd, dev_ln, with_skip=True)
        assert skip.data_ptr() == xd.data_ptr() and skip.grad_fn is not None
        loss = 0
        if use_y:
            yy = y.bfloat16() if bf16 else y
            loss = loss + (yy * gy.to(hip_device).to(yy.dtype)).sum()
        if use_s:
            loss = loss + (skip * gs.to(hip_device)).sum()
        loss.backward()
        assert (xd.grad.double().cpu() - xr.grad).abs().max() < (3e-2 if bf16 else 2e-5) * max(1.0, float(xr.grad.abs().max()))
        if use_y:
            assert (dev_ln.weight.grad.double().cpu() - ref_ln.weight.grad).abs().max() < 2e-4 * rows ** 0.5

