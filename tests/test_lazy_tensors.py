"""Host-side behaviour of the two lazily evaluated tensors of pika_amd.rnnt (no GPU: the kernels that fill them are
replaced by torch ops on CPU buffers).  What must hold: metadata never evaluates them, every aten op, autograd
accumulation, pickling and printing does, and autograd passes them between custom Functions untouched."""
import copy
import pickle

import torch

from pika_amd import rnnt


class _CpuState(rnnt.LogitsState):
    __slots__ = ("calls",)

    def __init__(self):
        super().__init__(1.0)
        self.calls = 0

    def to_log_probs(self, buf):
        if self.raw:
            self.calls += 1
            buf.copy_(torch.log_softmax(buf, -1))
            self.raw = False
        return buf


class _Compact(object):
    pass


def _lazy_grad(value):
    c = _Compact()
    c.ws, c.dims = torch.zeros(4, dtype=torch.uint8), tuple(value.shape) + (0,)
    g = rnnt.LazyDenseGrad(c, None, None, None)
    made = []
    g.dense = lambda: (made.append(1), value)[1] if g._dense is None and not made else value   # stands in for the writer kernel
    return g, made


def test_lazy_log_probs_metadata_is_free_and_every_op_sees_log_probs():
    torch.manual_seed(0)
    x = torch.randn(2, 3, 4, 8)
    st = _CpuState()
    lp = rnnt.LazyLogProbs(st, x.clone())
    assert (tuple(lp.shape), lp.dtype, lp.device.type, lp.dim(), lp.numel(), lp.is_contiguous()) == (
        (2, 3, 4, 8), torch.float32, "cpu", 4, 192, True)
    assert lp.size(-1) == 8 and not lp.is_cuda and st.raw and st.calls == 0
    want = torch.log_softmax(x, -1)
    assert torch.allclose(lp + 0.0, want) and st.calls == 1 and not st.raw  # first aten op normalises in place, once
    assert torch.allclose(lp.exp().sum(-1), torch.ones(2, 3, 4)) and st.calls == 1
    assert type(lp[0]) is torch.Tensor and type(lp.double()) is torch.Tensor
    assert torch.allclose(pickle.loads(pickle.dumps(lp)), want) and torch.allclose(copy.deepcopy(lp), want)
    assert "normalised=True" in repr(lp)


def test_lazy_tensors_travel_through_autograd_untouched_and_materialise_on_accumulation():
    torch.manual_seed(1)
    seen = []

    class Producer(torch.autograd.Function):     # stands in for JointOutFn
        @staticmethod
        def forward(ctx, x):
            ctx.st = _CpuState()
            return rnnt.LazyLogProbs(ctx.st, x * 1.0)

        @staticmethod
        def backward(ctx, g):
            seen.append(type(g))
            return g.dense() * 1.0 if isinstance(g, rnnt.LazyDenseGrad) else g

    class Loss(torch.autograd.Function):         # stands in for the RNN-T loss
        @staticmethod
        def forward(ctx, lp):
            assert isinstance(lp, rnnt.LazyLogProbs) and lp.requires_grad
            ctx.shape, ctx.raw = lp.shape, lp.state.raw
            return lp.buf.detach().sum() * 0.0 + 1.0

        @staticmethod
        def backward(ctx, go):
            g, _ = _lazy_grad(torch.full(tuple(ctx.shape), 3.0))
            return g

    x = torch.randn(2, 3, 4, 8, requires_grad=True)
    lp = Producer.apply(x)
    assert isinstance(lp, rnnt.LazyLogProbs) and lp.grad_fn is not None
    Loss.apply(lp).backward()
    assert seen == [rnnt.LazyDenseGrad] and torch.equal(x.grad, torch.full_like(x, 3.0))
    # a second consumer of log_probs: autograd adds the two gradients -> the lazy one is written, a plain tensor arrives
    seen.clear(); x.grad = None
    lp = Producer.apply(x)
    (Loss.apply(lp) + (lp * 2.0).sum()).backward()
    assert seen == [torch.Tensor] and torch.allclose(x.grad, torch.full_like(x, 5.0))
    # a leaf: .grad is a real tensor
    seen.clear()
    leaf = torch.zeros(2, 3, 4, 8, requires_grad=True)

    class LeafLoss(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            ctx.shape = t.shape
            return t.sum() * 0.0

        @staticmethod
        def backward(ctx, go):
            return _lazy_grad(torch.full(tuple(ctx.shape), 7.0))[0]

    LeafLoss.apply(leaf).backward()
    assert type(leaf.grad) is torch.Tensor and torch.equal(leaf.grad, torch.full_like(leaf, 7.0))
