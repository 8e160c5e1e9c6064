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


@pytest.mark.parametrize("M,V,H,scale", [(37, 104, 64, 1.0), (300, 5000, 128, 0.7), (5, 8, 64, 1.0)])
def test_joint_output_fn(hip_device, M, V, H, scale):
    """JointOutFn (fc2 + log-softmax, bf16 mode) vs the fp64 formula of transducer.py:107-111 on
    bf16-representable inputs.  Error sources left: fp32 accumulation/exp and the bf16 rounding of
    d(logits) in the backward (2^-9 relative per element)."""
    from pika_amd import gemm as G
    from pika_amd.model.hipops import JointOutFn, joint_out_ok
    old, G.PRECISION = G.PRECISION, "bf16"
    try:
        g = torch.Generator().manual_seed(M + V)
        h = (torch.randn(M, H, generator=g) * 0.5).bfloat16()
        w = (torch.randn(V, H, generator=g) * 0.2).bfloat16()
        b = torch.randn(V, generator=g) * 0.1
        gy = torch.randn(M, V, generator=g) * (torch.rand(M, V, generator=g) < 0.05)  # sparse like RNN-T
        h64, w64, b64 = [t.double().requires_grad_(True) for t in (h, w, b)]
        ref = torch.log_softmax(scale * (h64 @ w64.t() + b64), dim=-1)
        (ref * gy.double()).sum().backward()
        hd = h.to(hip_device).requires_grad_(True)
        wd = w.float().to(hip_device).requires_grad_(True)
        bd = b.to(hip_device).requires_grad_(True)
        assert joint_out_ok(hd, wd)
        out = JointOutFn.apply(hd, wd, bd, scale)
        assert (out.double().cpu() - ref.detach()).abs().max() < 2e-5
        (out * gy.to(hip_device)).sum().backward()
        for got, want in ((hd.grad, h64.grad), (wd.grad, w64.grad), (bd.grad, b64.grad)):
            s = want.abs().max().item()
            assert (got.double().cpu() - want).abs().max() < 1e-2 * s, (got.shape, s)
            # unbiased rounding: the mean error is far below the per-element bound
            assert abs((got.double().cpu() - want).mean().item()) < 1e-3 * s
    finally:
        G.PRECISION = old


@pytest.mark.parametrize("V", [40, 5000, 4616, 6268, 8192])
def test_joint_backward_uses_compact_rnnt_gradient(hip_device, V):
    """(V = 5000 / 4616 take the 8-column d(logits) kernel, 40 the 4-column one; 6268 -- the shipped recipes' vocabulary,
    not a multiple of 8: the last granule is masked -- and 8192 take its 13- and 16-block instantiations.)
    log_probs from JointOutFn straight into the RNN-T loss: the joint backward recognises the loss' own
    dense gradient tensor and rebuilds d(logits) from the two non-zeros per row kept in the loss workspace
    (pika_rnnt_dlogits_compact_bf16) -- same parameter gradients as the dense path; any tensor that is
    not that exact gradient (here: scaled by a hook) takes the dense path."""
    from pika_amd import gemm as G
    from pika_amd.model.hipops import JointOutFn
    from pika_amd.rnnt import RNNTLoss
    old, G.PRECISION = G.PRECISION, "bf16"
    try:
        g = torch.Generator().manual_seed(3)
        B, T, U, H = 2, 11, 4, 64
        h = (torch.randn(B, T, U + 1, H, generator=g) * 0.5).bfloat16().to(hip_device)
        w = (torch.randn(V, H, generator=g) * 0.3).to(hip_device)
        b = (torch.randn(V, generator=g) * 0.1).to(hip_device)
        labels = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32).to(hip_device)
        tl = torch.tensor([T, T - 3], dtype=torch.int32, device=hip_device)
        ul = torch.tensor([U, U - 1], dtype=torch.int32, device=hip_device)

        def run(hook):
            hh = h.clone().requires_grad_(True)
            ww = w.clone().requires_grad_(True)
            bb = b.clone().requires_grad_(True)
            lp = JointOutFn.apply(hh, ww, bb, 1.0)
            if hook:
                lp.register_hook(lambda t: t * 1.0)     # a NEW tensor reaches the joint backward
            RNNTLoss().apply(lp, labels, tl, ul).sum().backward()
            return hh.grad.float(), ww.grad, bb.grad
        before = JointOutFn.compact_hits
        compact = run(False)
        assert JointOutFn.compact_hits == before + 1
        dense = run(True)
        assert JointOutFn.compact_hits == before + 1
        for a, d in zip(compact[:2], dense[:2]):
            assert torch.allclose(a, d, rtol=1e-5, atol=1e-6 * d.abs().max().item())
        # the bias gradient of the compact path is summed inside the d(logits) kernel BEFORE the bf16 rounding
        # (the dense path sums the rounded matrix): agreement to the rounding of the summands, and closer to fp64
        a, d = compact[2], dense[2]
        assert (a - d).abs().max().item() < 3e-3 * d.abs().max().item()
    finally:
        G.PRECISION = old


def test_lazy_log_probs_and_lazy_rnnt_gradient(hip_device, monkeypatch):
    """pika_amd.rnnt.LazyLogProbs / LazyDenseGrad: the joint hands the loss its raw logits (the log-softmax pass and
    the dense gradient are only produced if something else needs them) -- same loss and parameter gradients as the
    eager chain log-softmax -> loss -> dense gradient -> log-softmax backward.  A hook on log_probs, a second consumer
    or retain_grad() make autograd / aten touch the tensors: they are then produced by the same kernels the eager
    path runs, and the values it sees are the eager ones."""
    from pika_amd import gemm as G
    from pika_amd.model.hipops import JointOutFn
    from pika_amd.rnnt import RNNTLoss, LazyDenseGrad, LazyLogProbs
    old, G.PRECISION = G.PRECISION, "bf16"
    try:
        g = torch.Generator().manual_seed(5)
        B, T, U, H, V = 3, 9, 5, 64, 48
        h = (torch.randn(B, T, U + 1, H, generator=g) * 0.5).bfloat16().to(hip_device)
        w = (torch.randn(V, H, generator=g) * 0.3).to(hip_device)
        b = (torch.randn(V, generator=g) * 0.1).to(hip_device)
        labels = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32).to(hip_device)
        tl = torch.tensor([T, T - 2, T - 4], dtype=torch.int32, device=hip_device)
        ul = torch.tensor([U, U - 1, U - 3], dtype=torch.int32, device=hip_device)
        scale = torch.tensor([1.0, 0.5, 2.0], device=hip_device)          # per-utterance grad_output, as in the MBR script

        def run(lazy, mode):
            monkeypatch.setenv("PIKA_RNNT_LAZY_GRAD", "1" if lazy else "0")
            hh, ww, bb = (t.clone().requires_grad_(True) for t in (h, w, b))
            lp = JointOutFn.apply(hh, ww, bb, 1.0, lazy)                   # what pika_amd.model.ops.joint does
            lp._pika_lazy_grad_ok = True
            assert isinstance(lp, LazyLogProbs) == lazy
            seen = []
            if mode == "hook":
                lp.register_hook(lambda t: (seen.append(type(t)), t * 1.0)[1])
            if mode == "retain":
                lp.retain_grad()
            pre = (lp.detach() * 1.0).sum() if mode == "read_first" else None   # normalises the buffer BEFORE the loss
            costs = RNNTLoss().apply(lp, labels, tl, ul)
            loss = (costs * scale).sum()
            if mode == "read_first":
                assert torch.isfinite(pre) and not (lazy and lp.state.raw)
            if mode == "second":
                loss = loss + 0.25 * (lp * lp).sum()
            raw_after_forward = lazy and lp.state.raw
            if mode == "twice":                                            # gradients accumulate: 2 x
                loss.backward(retain_graph=True)
            loss.backward()
            lpv = lp.detach().clone() if mode == "values" else None        # any aten op sees real log-probs
            return dict(dh=hh.grad.float(), dw=ww.grad, db=bb.grad, lpg=lp.grad if mode == "retain" else None,
                        seen=seen, costs=costs.detach(), raw=raw_after_forward, lp=lpv)

        def close(a, d, tol):
            assert (a - d).abs().max().item() <= tol * d.abs().max().item() + 1e-12

        hits = JointOutFn.compact_hits
        eager, lazy = run(False, "plain"), run(True, "plain")
        assert JointOutFn.compact_hits == hits + 2                        # both built d(logits) from the loss workspace
        assert lazy["raw"]                                                # the log-softmax pass never ran
        close(lazy["costs"], eager["costs"], 2e-6)
        close(lazy["dw"], eager["dw"], 2e-4)
        close(lazy["db"], eager["db"], 2e-4)
        close(lazy["dh"], eager["dh"], 1e-2)                              # bf16 matrices: a few one-ulp flips
        for mode in ("hook", "second", "retain", "values", "twice", "read_first"):
            e, l = run(False, mode), run(True, mode)
            close(l["costs"], e["costs"], 2e-6)
            close(l["dw"], e["dw"], 2e-4)
            close(l["db"], e["db"], 3e-3)                                 # summed before / after the bf16 rounding
            close(l["dh"], e["dh"], 1e-2)
            if mode == "hook":
                assert l["seen"] == [LazyDenseGrad] and e["seen"] == [torch.Tensor]
            if mode == "twice":
                close(l["dw"], 2.0 * lazy["dw"], 2e-4)
            if mode == "second":
                assert not l["raw"]                                       # lp * lp normalised the buffer before the loss
            if mode == "retain":
                assert type(l["lpg"]) is torch.Tensor
                close(l["lpg"], e["lpg"], 2e-6)
            if mode == "values":
                assert type(l["lp"]) is torch.Tensor
                close(l["lp"], e["lp"], 1e-6)
                assert torch.allclose(l["lp"].exp().sum(-1), torch.ones_like(l["lp"][..., 0]), atol=1e-5)
    finally:
        G.PRECISION = old


def test_joint_beyond_the_lazy_range(hip_device, monkeypatch):
    """ADVICE r1 (rnnt.py:232): vocabularies the lazy joint output cannot serve (V = 8200 > 8192: the d(logits)
    kernels hold one 64-padded row per wave; round 6 widened them from 5120 for the recipes' V = 6268) must take the plain chain -- linear, log-softmax, loss -- end to end,
    and a LazyLogProbs that reaches the loss already normalised (read first) must be handed over as log-probs."""
    import torch.nn as nn
    from pika_amd import gemm as G
    from pika_amd.model import ops
    from pika_amd.model.hipops import joint_out_ok
    from pika_amd.rnnt import RNNTLoss, LazyLogProbs
    old, G.PRECISION = G.PRECISION, "bf16"
    try:
        g = torch.Generator().manual_seed(11)
        B, T, U, H = 2, 5, 3, 64
        # (5001 / 6269: not multiples of four -- the joint pads the output layer to the next one and the lazy output stands for
        #  the layer's own columns; 8190: its padding would reach 8192)
        for V, lazy_expected in ((8200, False), (8192, True), (6268, True), (5001, True), (6269, True), (8190, True)):
            fc1, fcg, fc2 = nn.Linear(2 * H, H), nn.Linear(2 * H, H), nn.Linear(H, V)
            for m in (fc1, fcg, fc2):
                m.to(hip_device)
            enc = torch.randn(B, T, H, generator=g).to(hip_device)
            pred = torch.randn(B, U + 1, H, generator=g).to(hip_device)
            labels = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32).to(hip_device)
            tl = torch.tensor([T, T - 1], dtype=torch.int32, device=hip_device)
            ul = torch.tensor([U, U - 2], dtype=torch.int32, device=hip_device)
            assert joint_out_ok(torch.empty(1, H, dtype=torch.bfloat16, device=hip_device), fc2.weight,
                                units=(V + 3) & ~3) == lazy_expected
            # (with_labels: the 16-bit lattice of JointOutFn's labelled form, as Net.forward calls the joint)
            for read_first, with_labels in ((False, False), (True, False)) + (((False, True),) if V % 4 else ()):
                for m in (fc1, fcg, fc2):
                    m.zero_grad()
                lp = ops.joint(enc, pred, fc1, fcg, fc2, log_softmax=True, labels=labels.long() if with_labels else None)
                assert isinstance(lp, LazyLogProbs) == lazy_expected and lp.shape[-1] == V
                if with_labels:
                    assert lp.buf.dtype == torch.float16 and lp.buf.shape[-1] == (V + 3) & ~3
                if read_first:
                    assert torch.allclose(lp.detach().exp().sum(-1), torch.ones(B, T, U + 1, device=hip_device), atol=1e-4)
                costs = RNNTLoss().apply(lp, labels, tl, ul)
                costs.sum().backward()
                got = (costs.detach().double().cpu(), fc2.weight.grad.double().cpu(), fc1.weight.grad.double().cpu())
                # fp64 reference of the same chain
                z = torch.cat((enc.unsqueeze(2).expand(-1, -1, U + 1, -1), pred.unsqueeze(1).expand(-1, T, -1, -1)), -1).double()
                w = [m.weight.detach().double().requires_grad_(True) for m in (fc1, fcg, fc2)]
                bb = [m.bias.detach().double() for m in (fc1, fcg, fc2)]
                hh = torch.tanh(z @ w[0].t() + bb[0]) * torch.sigmoid(z @ w[1].t() + bb[1])
                lp64 = torch.log_softmax(hh @ w[2].t() + bb[2], -1)
                import numpy as np
                from oracle import rnnt as O
                c64, g64 = O.rnnt_loss(lp64.detach().float().cpu().numpy(), labels.cpu().numpy(), tl.cpu().numpy(), ul.cpu().numpy())
                lp64.backward(torch.from_numpy(g64).to(hip_device))
                assert np.allclose(got[0].numpy(), c64, rtol=3e-2)              # bf16 operands in every product
                for a, r in ((got[1], w[2].grad.cpu()), (got[2], w[0].grad.cpu())):
                    assert (a - r).norm() <= 6e-2 * r.norm(), (V, read_first, with_labels, float((a - r).norm() / r.norm()))
    finally:
        G.PRECISION = old


@pytest.mark.parametrize("V", [1000, 5000, 264, 6268])
def test_log_sum_exp_from_the_output_gemm_epilogue(hip_device, monkeypatch, V):
    """The joint's output GEMM emits per-row partial (max, sum exp) pairs per 64-column block in its epilogue and the
    loss merges them (pika_gemm_bf16_nt_lse -> pika_rnnt_fused_forward_partials) instead of re-reading the lattice of
    logits: same costs and parameter gradients as the path that reads the logits for the log-sum-exp."""
    from pika_amd import gemm as G
    from pika_amd.model import hipops as hipops_mod
    from pika_amd.model.hipops import JointOutFn
    from pika_amd.rnnt import RNNTLoss, LazyLogProbs
    old, G.PRECISION = G.PRECISION, "bf16"
    try:
        g = torch.Generator().manual_seed(V)
        B, T, U, H = 3, 37, 6, 128
        h = (torch.randn(B, T, U + 1, H, generator=g) * 0.5).bfloat16().to(hip_device)
        w = (torch.randn(V, H, generator=g) * 0.3).to(hip_device)
        b = (torch.randn(V, generator=g) * 2.0).to(hip_device)
        labels = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32).to(hip_device)
        tl = torch.tensor([T, T - 5, T - 11], dtype=torch.int32, device=hip_device)
        ul = torch.tensor([U, U - 1, U - 4], dtype=torch.int32, device=hip_device)

        def run(epi):
            monkeypatch.setattr(hipops_mod, "JOINT_LSE_EPILOGUE", bool(epi))
            hh, ww, bb = (t.clone().requires_grad_(True) for t in (h, w, b))
            lp = JointOutFn.apply(hh, ww, bb, 1.0, True)
            lp._pika_lazy_grad_ok = True
            assert isinstance(lp, LazyLogProbs) and (lp.state.partials is not None) == epi
            costs = RNNTLoss().apply(lp, labels, tl, ul)
            assert lp.state.raw
            costs.sum().backward()
            return costs.detach(), hh.grad.float(), ww.grad, bb.grad, lp.detach().clone()
        a, r = run(True), run(False)
        assert torch.allclose(a[0], r[0], rtol=2e-6, atol=1e-5)
        for x, y, tol in ((a[1], r[1], 1e-2), (a[2], r[2], 2e-4), (a[3], r[3], 2e-4)):
            assert (x - y).abs().max().item() <= tol * y.abs().max().item() + 1e-12
        assert torch.allclose(a[4], r[4], atol=1e-5)          # reading the values normalises both the same way
    finally:
        G.PRECISION = old


@pytest.mark.parametrize("V", [5000, 1000, 264, 6268, 4612])
def test_sixteen_bit_logits_keep_the_loss_in_fp32(hip_device, V):
    """JointOutFn with the lattice's labels (pika_gemm_bf16_nt_lse_f16): the (B,T,U1,V) logits exist only as fp16, yet
    * costs equal those of the fp32-logits path to fp32 rounding: log-sum-exp partials and the two logits per row the loss
      reads leave the product's epilogue in fp32 (rows beyond an utterance's lattice included);
    * d(hidden), d(weight), d(bias) agree with the fp32-logits path at the level of the bf16 rounding of d(logits) (the
      softmax of the backward is taken from the fp16 copy);
    * a loss given OTHER labels than the joint (or another blank) still gets the right cells -- from the fp16 copy;
    * reading the values returns the fp32 log-probabilities (the product runs again), and the backward still works."""
    from pika_amd import gemm as G
    from pika_amd.model.hipops import JointOutFn
    from pika_amd.rnnt import RNNTLoss, LazyLogProbs
    old, G.PRECISION = G.PRECISION, "bf16"
    try:
        g = torch.Generator().manual_seed(V + 1)
        B, T, U, H = 3, 37, 6, 128
        h = (torch.randn(B, T, U + 1, H, generator=g) * 0.5).bfloat16().to(hip_device)
        w = (torch.randn(V, H, generator=g) * 0.3).to(hip_device)
        b = (torch.randn(V, generator=g) * 2.0).to(hip_device)
        labels = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32).to(hip_device)
        other = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32).to(hip_device)
        tl = torch.tensor([T, T - 5, T - 11], dtype=torch.int32, device=hip_device)
        ul = torch.tensor([U, U - 1, U - 4], dtype=torch.int32, device=hip_device)

        def run(joint_labels, loss_labels=labels, blank=0, read=False):
            hh, ww, bb = (t.clone().requires_grad_(True) for t in (h, w, b))
            lp = JointOutFn.apply(hh, ww, bb, 1.0, True, joint_labels)
            lp._pika_lazy_grad_ok = True
            assert isinstance(lp, LazyLogProbs) and (lp.buf.dtype == torch.float16) == (joint_labels is not None)
            vals = lp.detach().clone() if read else None
            costs = RNNTLoss(blank=blank).apply(lp, loss_labels, tl, ul)
            costs.sum().backward()
            return costs.detach(), hh.grad.float(), ww.grad, bb.grad, vals
        ref, got = run(None), run(labels.long())
        assert torch.allclose(got[0], ref[0], rtol=2e-6, atol=1e-5), (got[0], ref[0])
        worst = [float((x - y).abs().max() / y.abs().max()) for x, y in zip(got[1:4], ref[1:4])]
        print("V=%d: 16-bit logits: cost difference %.1e, d(hidden) / d(weight) / d(bias) %.1e / %.1e / %.1e of the largest "
              "entry" % (V, float((got[0] - ref[0]).abs().max()), *worst))
        assert worst[0] < 1.5e-2 and worst[1] < 2e-3 and worst[2] < 2e-3, worst
        # other labels / another blank: the cells come from the fp16 copy (|logit| < 8 here: 2^-11 * 8 = 4e-3 each,
        # ~40 cells on a path)
        for kw in (dict(loss_labels=other), dict(blank=3)):
            r2, g2 = run(None, **kw), run(labels.long(), **kw)
            assert torch.allclose(g2[0], r2[0], rtol=2e-4, atol=5e-2), (kw.keys(), g2[0], r2[0])
        # a reader gets fp32 log-probabilities; loss and backward still work (dense path)
        rr, gr = run(None, read=True), run(labels.long(), read=True)
        assert torch.allclose(gr[4], rr[4], atol=1e-5)
        assert torch.allclose(gr[0], rr[0], rtol=2e-6, atol=1e-5)
        for x, y in zip(gr[1:4], rr[1:4]):
            assert (x - y).abs().max().item() <= 1e-2 * y.abs().max().item() + 1e-12
    finally:
        G.PRECISION = old


@pytest.mark.gpu
def test_sixteen_bit_logits_backward_takes_blank_and_label_logits_in_fp32(hip_device):
    """with |logit| ~ 30-60 the fp16 copy of the lattice is 2^-11 |logit| = 1.5-3e-2 off, i.e. percents of a
    softmax value -- and the blank / label entries of a row carry the loss' own gradient terms.  Their softmax comes from
    the fp32 pair the forward product kept (`gathered`): d(bias) of the blank column -- the sum of that entry over all
    lattice rows -- agrees with the fp32-logits path several times closer than with the fp16 values alone (measured with
    the fix-up switched off through a loss that is handed another blank, for which the pair does not apply)."""
    from pika_amd import gemm as G
    from pika_amd.model.hipops import JointOutFn
    from pika_amd.rnnt import RNNTLoss
    old, G.PRECISION = G.PRECISION, "bf16"
    try:
        V = 5000
        g = torch.Generator().manual_seed(77)
        B, T, U, H = 3, 37, 6, 128
        h = (torch.randn(B, T, U + 1, H, generator=g) * 1.5).bfloat16().to(hip_device)
        w = (torch.randn(V, H, generator=g) * 0.6).to(hip_device)          # |logit| up to ~60
        b = (torch.randn(V, generator=g) * 2.0).to(hip_device)
        b[0] += 30.0                                                         # a blank that takes a share of every row
        labels = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32).to(hip_device)
        tl = torch.tensor([T, T - 5, T - 11], dtype=torch.int32, device=hip_device)
        ul = torch.tensor([U, U - 1, U - 4], dtype=torch.int32, device=hip_device)

        def run(joint_labels):
            hh, ww, bb = (t.clone().requires_grad_(True) for t in (h, w, b))
            lp = JointOutFn.apply(hh, ww, bb, 1.0, True, joint_labels)
            lp._pika_lazy_grad_ok = True
            RNNTLoss(blank=0).apply(lp, labels, tl, ul).sum().backward()
            return bb.grad, float(lp.buf.float().abs().max())
        ref, _ = run(None)
        got, big = run(labels.long())
        assert big > 25.0, big
        cols = torch.cat([torch.zeros(1, dtype=torch.long, device=hip_device), labels.long().unique()])
        err = ((got - ref)[cols].abs() / ref[cols].abs().clamp(min=1e-3)).max().item()
        print("largest |logit| %.0f: d(bias) of the blank / label columns, 16-bit logits vs fp32 logits: %.2e" % (big, err))
        assert err < 1e-2, err      # (bf16 rounding of the entries themselves is 4e-3; the fp16 softmax alone 1-3e-2)
    finally:
        G.PRECISION = old
