"""Host side of the RNN-T loss: mirrors the `warp_rnnt` surface PIKA imports.

Reference call site (trainer/train_transducer_bmuf_otfaug.py:58,97-99):

    transducer_loss = RNNTLoss(blank=0, reduction='sum').apply
    loss = transducer_loss(outputs, target_batch.int(), len_batch, ali_lens)
    loss = loss.sum()

so `RNNTLoss(...)` must be constructible with keyword arguments and expose `.apply(log_probs,
labels, frames_lengths, labels_lengths)` returning per-utterance costs (B,) that the caller
sums (and in the MBR script pre-multiplies by a float, train_transducer_mbr_bmuf_otfaug.py:157).

MI355X-first split: forward runs only gather + alpha/beta (~0.1 ms) and keeps the lattice in a
34 MB workspace; the dense (B,T,U1,V) gradient is written ONCE, in backward, already scaled
by autograd's grad_output -- warp_rnnt instead materialises it in forward and multiplies it
again in backward (3 extra passes over 32 GB at the benchmark shape).  When log_probs came out of
this package's own joint network, the gradient goes back as a LazyDenseGrad: the joint's backward
reads the loss workspace and the dense tensor is written only if anything else touches it.
"""
import torch

from . import _lib


# bench.py hook: when set to {"fwd": [], "bwd": []}, every C-ABI call is bracketed by HIP events
# recorded on the launch stream (torch's current stream) so kernel time is measured live.
KERNEL_EVENTS = None

# widest vocabulary the fused lattice kernels take (include/pika_rnnt.h: one wave covers a row in 64 x 4 x 32 columns; the
# benchmarked V = 5000 runs the 20-register instantiation, the recipes' 6268 the 32-register one)
MAX_FUSED_V = 8192


class _timed(object):
    def __init__(self, key):
        self.key = key

    def __enter__(self):
        if KERNEL_EVENTS is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if KERNEL_EVENTS is not None:
            self.e1.record()
            KERNEL_EVENTS[self.key].append((self.e0, self.e1))


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check_inputs(log_probs, labels, frames_lengths, labels_lengths, blank):
    # same argument checks, same exception types as the reference binding's Python wrapper
    if not log_probs.is_cuda:
        raise RuntimeError("pika_amd RNNTLoss: log_probs must live on a HIP device "
                           "(there is no CPU path; the CPU checker lives in oracle/ for tests only)")
    if log_probs.dtype != torch.float32:
        raise TypeError("log_probs must be float32, got %s" % log_probs.dtype)
    for name, t in (("labels", labels), ("frames_lengths", frames_lengths),
                    ("labels_lengths", labels_lengths)):
        if t.dtype != torch.int32:
            raise TypeError("%s must be int32, got %s" % (name, t.dtype))
        if t.device != log_probs.device:
            raise RuntimeError("%s is on %s but log_probs is on %s" % (name, t.device, log_probs.device))
    if log_probs.dim() != 4:
        raise ValueError("log_probs must be (B,T,U+1,V), got %s" % (tuple(log_probs.shape),))
    B, T, U1, V = log_probs.shape
    if labels.dim() != 2 or labels.shape[0] != B or labels.shape[1] != U1 - 1:
        raise ValueError("labels must be (B,U)=(%d,%d), got %s" % (B, U1 - 1, tuple(labels.shape)))
    if frames_lengths.shape != (B,) or labels_lengths.shape != (B,):
        raise ValueError("frames_lengths / labels_lengths must be (B,)")
    if not 0 <= blank < V:
        raise ValueError("blank=%d outside [0,%d)" % (blank, V))
    if U1 > 1024:
        raise ValueError("U+1=%d > 1024 not supported" % U1)


class CompactGrad(object):
    """What the backward leaves on the dense gradient tensor it returns (`grads._pika_compact`): the
    workspace whose row metadata hold the (at most two) non-zeros of every V-row.  A consumer that
    produced log_probs itself (pika_amd.model.hipops.JointOutFn) may use it instead of reading the
    7.8 GB dense tensor back -- only if the tensor it received IS this tensor, unmodified (`matches`)."""

    __slots__ = ("ws", "dims", "ptr", "version")

    def __init__(self, ws, dims, grads):
        self.ws, self.dims, self.ptr, self.version = ws, dims, grads.data_ptr(), grads._version

    def matches(self, g):
        return g.data_ptr() == self.ptr and g._version == self.version and tuple(g.shape) == tuple(self.dims[:4])


def _lazy_enabled():
    import os
    return os.environ.get("PIKA_RNNT_LAZY_GRAD", "1") != "0"


class LazyDenseGrad(torch.Tensor):
    """The loss' (B,T,U1,V) gradient as a tensor that is only written when somebody looks at it.

    The backward of the loss always leaves the at most two non-zeros of every V-row in its workspace.  A
    producer that computed log_probs itself and registered for it (`log_probs._pika_lazy_grad_ok = True`, set
    by pika_amd.model.ops.joint on the output of JointOutFn) gets this object instead of 7.8 GB of mostly zeros;
    JointOutFn.backward recognises it and builds d(logits) from the workspace (`compact`).  ANY other use -- an
    aten op on it, autograd accumulating a second gradient into log_probs, a hook, `.grad` of a leaf -- goes
    through __torch_dispatch__, which first writes the dense tensor (`pika_rnnt_loss_dense_grads`, the very
    streaming pass the eager path runs) and then runs the op on it: values are identical in every case."""

    @staticmethod
    def __new__(cls, compact, labels, frames_lengths, labels_lengths, width=None):
        # width: the vocabulary the CALLER sees when the kernels run on a wider one (an output layer padded to a multiple of
        # four units inside the joint, pika_amd.model.ops.joint): the tensor this object stands for has `width` columns
        B, T, U1, V, _ = compact.dims
        r = torch.Tensor._make_wrapper_subclass(cls, (B, T, U1, V if width is None else int(width)), dtype=torch.float32,
                                                device=compact.ws.device, requires_grad=False)
        r.compact = compact
        r._keep = (labels, frames_lengths, labels_lengths)   # the metadata kernel has run; kept for symmetry of lifetimes
        r._dense = None
        r.lse = None
        return r

    def dense(self):
        if self._dense is None:
            B, T, U1, V, blank = self.compact.dims
            ws = self.compact.ws
            with torch.cuda.device(ws.device):
                g = torch.empty((B, T, U1, V), dtype=torch.float32, device=ws.device)
                with _timed("bwd"):
                    _lib.check(_lib.lib().pika_rnnt_loss_dense_grads(_ptr(ws), B, T, U1, V, blank, _ptr(g), _stream()),
                               "pika_rnnt_loss_dense_grads")
            self._dense = g if V == self.shape[-1] else g[..., :self.shape[-1]].contiguous()
        return self._dense

    def __repr__(self):
        return "LazyDenseGrad(shape=%s, written=%s)" % (tuple(self.shape), self._dense is not None)

    def __reduce_ex__(self, proto):
        return self.dense().__reduce_ex__(proto)

    def __deepcopy__(self, memo):
        return self.dense().clone()

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map
        un = lambda t: t.dense() if isinstance(t, LazyDenseGrad) else t   # noqa: E731
        return func(*tree_map(un, args), **tree_map(un, kwargs or {}))


class LogitsState(object):
    """Shared by a LazyLogProbs and the joint node that made it: does the (B,T,U1,V) buffer still hold raw logits?
    (Small on purpose: autograd nodes outlive their backward for as long as anything holds the graph, so nothing that
    hangs off them may keep the 7.8 GB buffer alive -- the buffer itself lives in autograd's saved tensors and in the
    LazyLogProbs the caller holds.)"""

    __slots__ = ("scale", "raw", "partials", "gathered", "recompute", "dense_lp")

    def __init__(self, scale):
        self.scale, self.raw = float(scale), True
        self.partials = None      # (2, rows, n_part) per-row partial (max, sum exp) pairs from the GEMM epilogue
        # 16-bit logits (pika_gemm_bf16_nt_lse_f16): the buffer is an fp16 matrix; `gathered` = (values (rows, 2) f32,
        # labels (B, U) i32, blank column) holds what the loss reads in fp32; `recompute()` runs the product again with an
        # fp32 output and returns log_softmax of it -- what ANY reader other than this package's loss gets (`dense_lp`)
        self.gathered = None
        self.recompute = None
        self.dense_lp = None

    def to_log_probs(self, buf):
        """The log-probabilities: in place in an fp32 buffer (buf <- log_softmax(scale * buf), once), or -- 16-bit logits --
        a separate fp32 tensor made by running the product again."""
        if self.recompute is not None:
            if self.dense_lp is None:
                self.dense_lp = self.recompute()
                self.raw = False
                self.partials = None
            return self.dense_lp
        if self.raw:
            B, T, U1, V = buf.shape
            with torch.cuda.device(buf.device):
                _lib.check(_lib.lib().pika_log_softmax_rows(buf.data_ptr(), B * T * U1, V, V, self.scale, _stream()),
                           "pika_log_softmax_rows")
            self.raw = False
            self.partials = None
        return buf


class LazyLogProbs(torch.Tensor):
    """log_softmax(logits) over the lattice as a tensor whose log-softmax pass only runs if somebody needs the
    values.  pika_amd.model.hipops.JointOutFn returns it; this module's loss takes the row log-sum-exp and the two
    log-probs per lattice cell it needs in ONE read of the raw logits (pika_rnnt_fused_forward), and the joint's
    backward subtracts that log-sum-exp on the fly for as long as the buffer is raw.  ANY other use -- an aten op,
    printing, `.float()`, a different loss -- goes through __torch_dispatch__, which first normalises the buffer in
    place (the same kernel the eager path runs in the forward) and then runs the op on the real log-probabilities."""

    @staticmethod
    def __new__(cls, state, buf, width=None):
        # (buf: the fp32 logits, or the fp16 matrix of a 16-bit joint; the tensor this object stands for is fp32 either way.
        #  width: the caller's vocabulary when the joint padded its output layer to a multiple of four units -- the extra
        #  columns hold logits of -6e4, probability zero: the tensor this object stands for has `width` columns)
        shape = tuple(buf.shape) if width is None else tuple(buf.shape[:-1]) + (int(width),)
        r = torch.Tensor._make_wrapper_subclass(cls, shape, dtype=torch.float32, device=buf.device, requires_grad=False)
        r.state, r.buf = state, buf
        return r

    def dense(self):
        full = self.state.to_log_probs(self.buf)
        return full if full.shape[-1] == self.shape[-1] else full[..., :self.shape[-1]]

    def __repr__(self):
        return "LazyLogProbs(shape=%s, normalised=%s)" % (tuple(self.shape), not self.state.raw)

    def __reduce_ex__(self, proto):     # pickling / torch.save: the real log-probabilities
        return self.dense().__reduce_ex__(proto)

    def __deepcopy__(self, memo):
        return self.dense().clone()

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map
        un = lambda t: t.dense() if isinstance(t, LazyLogProbs) else t   # noqa: E731
        return func(*tree_map(un, args), **tree_map(un, kwargs or {}))


class _RNNTLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_probs, labels, frames_lengths, labels_lengths, blank=0):
        wide = getattr(log_probs, "_pika_labels", None)
        if wide is not None and labels.dim() == 2 and log_probs.dim() == 4 and labels.shape[1] < log_probs.shape[2] - 1 \
                and wide.shape == (labels.shape[0], log_probs.shape[2] - 1):
            # a replayed forward (pika_amd/train_graph.py) padded the label axis to its bucket: its own copy of the labels
            # over that axis (same values in the caller's columns; nothing beyond labels_lengths is ever read)
            labels = wide
        _check_inputs(log_probs, labels, frames_lengths, labels_lengths, blank)
        ctx.lazy = bool(getattr(log_probs, "_pika_lazy_grad_ok", False)) and _lazy_enabled()
        lib = _lib.lib()
        labels = labels.contiguous()
        frames_lengths = frames_lengths.contiguous()
        labels_lengths = labels_lengths.contiguous()
        B, T, U1, V = log_probs.shape
        lse = None
        state = log_probs.state if isinstance(log_probs, LazyLogProbs) else None
        ctx.width = None
        if state is not None:
            # (a joint that padded its output layer: the kernels see the buffer's columns, the caller `V` of them)
            Vk = log_probs.buf.shape[-1]
            ctx.width, V = (V, Vk) if Vk != V else (None, V)
        if state is not None and not (state.raw and ctx.lazy and state.scale == 1.0 and V % 4 == 0 and V <= MAX_FUSED_V
                                      and (state.gathered is None or state.partials is not None)):
            state, V = None, log_probs.shape[-1]
            ctx.width = None
        if state is not None:
            # raw logits of this package's joint: log-sum-exp + gather in one read, no log-prob tensor
            x = log_probs.buf
            with torch.cuda.device(x.device):
                costs = torch.empty(B, dtype=torch.float32, device=x.device)
                # a replayed forward (pika_amd/train_graph.py) names the buffers its captured backward reads
                ws, lse = getattr(log_probs, "_pika_loss_buffers", None) or (None, None)
                n_ws = lib.pika_rnnt_workspace_bytes(B, T, U1)
                if ws is None or ws.numel() != n_ws or lse.numel() != B * T * U1 or ws.device != x.device:
                    lse = torch.empty(B * T * U1, dtype=torch.float32, device=x.device)
                    ws = torch.empty(n_ws, dtype=torch.uint8, device=x.device)
                part = state.partials
                state.partials = None          # one use: 250 MB at the benchmark shape
                with _timed("fwd"):
                    if state.gathered is not None:
                        gath, g_labels, g_blank = state.gathered
                        _lib.check(lib.pika_rnnt_fused_forward_gathered(
                            _ptr(x), x.stride(-2), _ptr(gath), _ptr(g_labels), int(g_blank), part[0].data_ptr(),
                            part[1].data_ptr(), part.shape[2], _ptr(labels), _ptr(frames_lengths), _ptr(labels_lengths),
                            B, T, U1, V, blank, _ptr(costs), _ptr(lse), _ptr(ws), _stream()),
                            "pika_rnnt_fused_forward_gathered")
                    elif part is not None:
                        _lib.check(lib.pika_rnnt_fused_forward_partials(
                            _ptr(x), part[0].data_ptr(), part[1].data_ptr(), part.shape[2], _ptr(labels),
                            _ptr(frames_lengths), _ptr(labels_lengths), B, T, U1, V, blank, _ptr(costs), _ptr(lse),
                            _ptr(ws), _stream()), "pika_rnnt_fused_forward_partials")
                    else:
                        _lib.check(lib.pika_rnnt_fused_forward(
                            _ptr(x), _ptr(labels), _ptr(frames_lengths), _ptr(labels_lengths), B, T, U1, V, blank,
                            _ptr(costs), _ptr(lse), _ptr(ws), _stream()), "pika_rnnt_fused_forward")
        else:
            # a LazyLogProbs the fused path cannot take (already read, scaled, V out of the fused kernel's range) is
            # normalised HERE: `.contiguous()` on the wrapper subclass short-circuits and would hand back the wrapper
            lp = log_probs.dense().contiguous() if isinstance(log_probs, LazyLogProbs) else log_probs.contiguous()
            with torch.cuda.device(lp.device):
                costs = torch.empty(B, dtype=torch.float32, device=lp.device)
                ws = torch.empty(lib.pika_rnnt_workspace_bytes(B, T, U1), dtype=torch.uint8,
                                 device=lp.device)
                with _timed("fwd"):
                    _lib.check(lib.pika_rnnt_loss_forward(
                        _ptr(lp), _ptr(labels), _ptr(frames_lengths), _ptr(labels_lengths),
                        B, T, U1, V, blank, _ptr(costs), _ptr(ws), _stream()), "pika_rnnt_loss_forward")
        # lse: the log-sum-exp of every row of the RAW logits (None when log-probs were read); it only means something
        # to the joint's backward while its buffer is still raw, which that backward checks itself
        ctx.save_for_backward(labels, frames_lengths, labels_lengths, ws, lse)
        ctx.dims = (B, T, U1, V, blank)
        return costs

    @staticmethod
    def backward(ctx, grad_costs):
        labels, frames_lengths, labels_lengths, ws, lse = ctx.saved_tensors
        B, T, U1, V, blank = ctx.dims
        lib = _lib.lib()
        gc = grad_costs.to(torch.float32).contiguous()
        if ctx.lazy:
            with torch.cuda.device(ws.device):
                _lib.check(lib.pika_rnnt_loss_backward(
                    _ptr(labels), _ptr(frames_lengths), _ptr(labels_lengths), B, T, U1, V, blank,
                    _ptr(gc), _ptr(ws), None, _stream()), "pika_rnnt_loss_backward")
            compact = CompactGrad.__new__(CompactGrad)
            compact.ws, compact.dims, compact.ptr, compact.version = ws, (B, T, U1, V, blank), 0, 0
            lazy = LazyDenseGrad(compact, labels, frames_lengths, labels_lengths, width=ctx.width)
            lazy.lse = lse
            return lazy, None, None, None, None
        with torch.cuda.device(ws.device):
            grads = torch.empty((B, T, U1, V), dtype=torch.float32, device=ws.device)
            with _timed("bwd"):
                _lib.check(lib.pika_rnnt_loss_backward(
                    _ptr(labels), _ptr(frames_lengths), _ptr(labels_lengths), B, T, U1, V, blank,
                    _ptr(gc), _ptr(ws), _ptr(grads), _stream()), "pika_rnnt_loss_backward")
        grads._pika_compact = CompactGrad(ws, (B, T, U1, V, blank), grads)
        return grads, None, None, None, None


class _FusedLogitsLossFn(torch.autograd.Function):
    """RNN-T costs straight from the joint's RAW logits (SURVEY 8d M1', include/pika_rnnt.h): the log-softmax
    is folded into the two passes the loss makes anyway (log-sum-exp + gather; gradient), so the (B,T,U1,V)
    log-prob tensor and its dense gradient never exist: 3 tensor passes instead of 6."""

    @staticmethod
    def forward(ctx, logits, labels, frames_lengths, labels_lengths, blank=0):
        _check_inputs(logits, labels, frames_lengths, labels_lengths, blank)
        lib = _lib.lib()
        x = logits.contiguous()
        labels, frames_lengths, labels_lengths = (t.contiguous() for t in (labels, frames_lengths, labels_lengths))
        B, T, U1, V = x.shape
        with torch.cuda.device(x.device):
            costs = torch.empty(B, dtype=torch.float32, device=x.device)
            lse = torch.empty(B * T * U1, dtype=torch.float32, device=x.device)
            ws = torch.empty(lib.pika_rnnt_workspace_bytes(B, T, U1), dtype=torch.uint8, device=x.device)
            with _timed("fwd"):
                _lib.check(lib.pika_rnnt_fused_forward(
                    _ptr(x), _ptr(labels), _ptr(frames_lengths), _ptr(labels_lengths), B, T, U1, V, blank,
                    _ptr(costs), _ptr(lse), _ptr(ws), _stream()), "pika_rnnt_fused_forward")
        ctx.save_for_backward(x, labels, frames_lengths, labels_lengths, ws, lse)
        ctx.dims = (B, T, U1, V, blank)
        return costs

    @staticmethod
    def backward(ctx, grad_costs):
        x, labels, frames_lengths, labels_lengths, ws, lse = ctx.saved_tensors
        B, T, U1, V, blank = ctx.dims
        gc = grad_costs.to(torch.float32).contiguous()
        with torch.cuda.device(x.device):
            grads = torch.empty_like(x)
            with _timed("bwd"):
                _lib.check(_lib.lib().pika_rnnt_fused_backward(
                    _ptr(x), _ptr(lse), _ptr(labels), _ptr(frames_lengths), _ptr(labels_lengths), B, T, U1, V,
                    blank, _ptr(gc), _ptr(ws), _ptr(grads), 0, V, _stream()), "pika_rnnt_fused_backward")
        return grads, None, None, None, None


def rnnt_loss_from_logits(logits, labels, frames_lengths, labels_lengths, blank=0):
    """Per-utterance costs of log_softmax(logits) under the RNN-T loss, differentiable w.r.t. the logits,
    without materialising the log-probabilities (V % 4 == 0, V <= 8192)."""
    return _FusedLogitsLossFn.apply(logits, labels, frames_lengths, labels_lengths, blank)


def rnnt_loss(log_probs, labels, frames_lengths, labels_lengths, average_frames=False,
              reduction=None, blank=0):
    """Functional form (same keyword surface as warp_rnnt.rnnt_loss)."""
    costs = _RNNTLossFn.apply(log_probs, labels, frames_lengths, labels_lengths, blank)
    if average_frames:
        costs = costs / frames_lengths.to(costs)
    if reduction == "sum":
        return costs.sum()
    if reduction == "mean":
        return costs.mean()
    if reduction in (None, "none"):
        return costs
    raise ValueError("Unknown reduction: %r" % (reduction,))


class RNNTLoss(object):
    """`RNNTLoss(blank=0, reduction='sum').apply(...)` exactly as the reference scripts use it.

    As with the binding the reference imports, constructor keywords other than `blank` do not
    change what `.apply` returns: per-utterance costs (B,), which the caller reduces itself
    (train_transducer_bmuf_otfaug.py:99 `loss = loss.sum()`).
    """

    def __init__(self, blank=0, reduction="sum", **unused):
        self.blank = int(blank)
        self.reduction = reduction

    def apply(self, log_probs, labels, frames_lengths, labels_lengths):
        return _RNNTLossFn.apply(log_probs, labels, frames_lengths, labels_lengths, self.blank)

    __call__ = apply


def export_lattice(ctx_ws, frames_lengths, labels_lengths, B, T, U1):
    """Diagnostic: dense (B,T,U1) alpha/beta from a workspace tensor (tests only)."""
    lib = _lib.lib()
    a = torch.empty((B, T, U1), dtype=torch.float32, device=ctx_ws.device)
    b = torch.empty_like(a)
    _lib.check(lib.pika_rnnt_export_lattice(_ptr(ctx_ws), _ptr(frames_lengths), _ptr(labels_lengths),
                                            B, T, U1, _ptr(a), _ptr(b), _stream()),
               "pika_rnnt_export_lattice")
    return a, b
