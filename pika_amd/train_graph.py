"""The training step of the reference loop as hipGraph replays -- BEHIND `Net.forward`, so that the UNCHANGED training
script (trainer/train_transducer_bmuf_otfaug.py:95-110) gets it:

    outputs = model.forward(data_batch, target_batch, len_batch, True)     # replay of graph F (everything in forward)
    loss = transducer_loss(outputs, target_batch.int(), len_batch, ali_lens).sum()   # ~6 eager launches (this package's loss)
    loss.backward()                                                        # loss backward (1 launch) + replay of graph B
    torch.nn.utils.clip_grad_norm_(model.parameters(), args.grad_clip, norm_type=inf)   # 2 launches (pika_amd/optim.py)
    optimizer.step()                                                       # 1 launch; lr / momentum read per call

The eager step is ~650 kernel launches issued from Python (22 ms of host CPU for 45 ms of device time); with eight ranks
on one host that is where scaling goes.  Here the ~600 launches of the model's forward and backward are captured once
per batch shape, the way torch.cuda.make_graphed_callables splits a callable at its autograd boundary:

  * graph F: `Net.forward` on static copies of (data, labels, lengths), recorded with autograd on;
  * graph B: `torch.autograd.grad(outputs, parameters, grad_outputs=<the loss' compact gradient>)` of that recording;
  * `Net.forward` copies the batch into the static inputs, replays F and returns the static output buffer -- a fresh
    `LazyLogProbs` over the raw logits, whose grad_fn is `_GraphedFn`; this package's loss reads it eagerly (log-sum-exp
    partials from the GEMM epilogue + two gathered logits per lattice cell) into static workspace / lse buffers;
  * `_GraphedFn.backward` gets the loss' `LazyDenseGrad` (never written: the non-zeros sit in the workspace), replays B
    and hands the static gradient tensors to `p.grad` (accumulating if the caller left gradients in place).

What makes the sequence replayable: fixed shapes (one pair of graphs per (data shape, label-axis width); a batch whose label
axis is a few labels narrower, or whose time axis a few frames shorter, than an existing pair's is padded onto it -- the
encoder then counts the data frames only (BatchNorm statistics, attention keys: model/encoder.py); the graphs share ONE
memory pool, so the activations of all shapes occupy the same memory, and an LRU bound caps the static outputs -- the
logits alone are 7.8 GB at B = 32, T' = 240); dropout through a device-side salt word every seeded kernel of
libpika_amd.so adds to its seed (`pika_set_dropout_salt`), re-drawn on the device before every forward; nothing on the
path reads the device.  The optimizer is NOT captured: clip + Nesterov SGD are three eager multi-tensor launches that read
lr / momentum from the optimizer object the script rebuilds after every BMUF block (:115-123).

Whatever does not fit runs the eager step with the same values: the first `warmup` training calls (lazy workspaces,
kernel attributes and the allocator reach their steady state), evaluation / no-grad calls, `softmax=False`, packed
sequences (LSTM encoders), a shape seen fewer than `min_seen` times, a capture that fails (remembered; warned once).
What cannot be served raises instead of computing something else: a backward that does not belong to the latest forward,
log-probs that were READ between forward and backward (that normalises the static logits in place; set
PIKA_TRAIN_GRAPH=0 for such a loop), a gradient that is not this package's loss gradient.

Switches: PIKA_TRAIN_GRAPH=0 (environment) turns the graphs off; PIKA_TRAIN_GRAPH_DEBUG=1/2/3 prints statistics / every
call / where a non-finite value first appears.  Everything else is an argument of `enable()` or an entry of DEFAULTS
(module dictionary; what `pika_amd.launch` leaves untouched): warmup (2 eager calls first), max_graphs (16 shapes kept),
min_seen (2: a shape is captured the second time it appears after the warm-up -- a corpus whose batch lengths never recur
stays eager instead of capturing every step), u_bucket (8: a batch rides on graphs whose label axis is up to 7 labels wider
than its own, padded with the embedding's padding index), t_bucket (64: ... and whose time axis is up to 63 frames longer;
0: exact frame counts only), bucket_capture (True: a bucket of t_bucket frames x u_bucket labels that has shown two different shapes
gets a pair of graphs at its upper boundary -- corpora whose exact shapes never recur), freeze_salt (False; tests: keep the dropout
salt word as it is), debug_sync (0; bit mask of
host syncs around the replays: 1 before F, 2 after F, 4 before B, 8 after B -- profiles/r4_hip_graph_packet_capture.txt).
"""
import collections
import os
import warnings

import torch

from . import _lib

_SALT = {"word": None, "users": 0}
DEFAULTS = {"warmup": 2, "max_graphs": 16, "min_seen": 2, "u_bucket": 8, "t_bucket": 64, "freeze_salt": False, "debug_sync": 0,
            "bucket_capture": True}


def _salt_word(device):
    """The process-wide dropout salt word (one process drives one GPU): allocated once and never freed, so a registered
    pointer can never dangle; registered while anybody uses it."""
    w = _SALT["word"]
    if w is None:
        w = _SALT["word"] = torch.zeros(1, dtype=torch.int32, device=device)
    if w.device != torch.device(device):
        raise RuntimeError("pika_amd.train_graph: one process drives one GPU (salt word lives on %s, asked for %s)"
                           % (w.device, device))
    return w


def _salt_acquire(device):
    w = _salt_word(device)
    if _SALT["users"] == 0:
        _lib.check(_lib.lib().pika_set_dropout_salt(w.data_ptr()), "pika_set_dropout_salt")
    _SALT["users"] += 1
    return w


def _salt_release():
    if _SALT["users"] > 0:
        _SALT["users"] -= 1
        if _SALT["users"] == 0:
            _lib.lib().pika_set_dropout_salt(None)


class _Entry(object):
    __slots__ = ("key", "gf", "gb", "inputs", "labels32", "logits", "partials", "ws", "lse", "dims", "grads", "gen", "scale",
                 "kind", "gout", "gathered", "recompute", "t_valid")


class StepGraphs(object):
    """Per-model state of the graphed step (hangs off the module as `_step_graphs`; not pickled)."""

    def __init__(self, model, warmup=None, max_graphs=None, min_seen=None, t_bucket=None):
        D = DEFAULTS
        self.warmup = int(D["warmup"]) if warmup is None else int(warmup)
        self.max_graphs = max(1, int(D["max_graphs"]) if max_graphs is None else int(max_graphs))
        self.min_seen = max(1, int(D["min_seen"]) if min_seen is None else int(min_seen))
        self.u_bucket = max(1, int(D["u_bucket"]))
        # time axis: a batch of T frames rides on graphs captured for up to t_bucket - 1 more frames (0: exact T only)
        self.t_bucket = max(0, int(D["t_bucket"]) if t_bucket is None else int(t_bucket))
        self.entries = collections.OrderedDict()        # key -> _Entry, least recently used first
        self.seen = {}
        self.bucket_shapes = {}                         # bucket key -> the distinct (frames, labels) shapes it has shown
        self.used_at = {}                               # entry key -> st.calls of its last replay
        self.calls = 0
        self.pool = None
        self.param_ptrs = None
        self.params = None
        self.broken = None                              # reason the model's forward cannot be captured
        self.last = None                                # (entry, generation) of the latest graphed forward
        self.freeze_salt = bool(D["freeze_salt"])        # tests / debugging: keep the salt word as it is
        self.stats = {"replays": 0, "captures": 0, "eager": 0, "evictions": 0}
        import weakref
        self.model_ref = weakref.ref(model)
        self.salt = _salt_acquire(next(model.parameters()).device)
        self._closed = False

    def clear(self):
        self.entries.clear()
        self.used_at.clear()
        self.last = None

    def close(self):
        if not self._closed:
            self._closed = True
            self.clear()
            _salt_release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def enable(model, warmup=None, max_graphs=None, min_seen=None, t_bucket=None):
    """Turn the graphed step on for `model` (a pika_amd.model.transducer.Net).  Idempotent; returns the state object."""
    st = model.__dict__.get("_step_graphs")
    if st is None or st._closed:
        st = StepGraphs(model, warmup, max_graphs, min_seen, t_bucket)
        model.__dict__["_step_graphs"] = st
        if os.environ.get("PIKA_TRAIN_GRAPH_DEBUG"):
            import atexit
            atexit.register(lambda: print("train_graph stats: %s, shapes kept %d, broken: %s" % (
                st.stats, len(st.entries), st.broken), flush=True))
    return st


def disable(model):
    st = model.__dict__.pop("_step_graphs", None)
    if st is not None:
        st.close()


AUTO = False        # pika_amd.launch sets this for training scripts: every Net that trains on a HIP device is enabled


def wanted(model, x, softmax):
    """Does this forward call go through the graphs?  (The cheap checks; `forward` below does the rest.)"""
    if os.environ.get("PIKA_TRAIN_GRAPH", "1") == "0":
        return False
    st = model.__dict__.get("_step_graphs")
    if st is None:
        if not AUTO or not (model.training and x.is_cuda and torch.is_grad_enabled()):
            return False
        st = enable(model)
    if st.broken is None:
        import pika_amd
        if not pika_amd.HIP_GRAPHS_SAFE_TO_ALTERNATE:
            _give_up(st, "the HIP runtime was initialised with its graph packet-capture fast path on, under which alternating "
                         "graph replays go wrong (pika_amd/__init__.py): import pika_amd before the first HIP call, or "
                         "export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0")
    return (st.broken is None and not st._closed and model.training and softmax and x.is_cuda and torch.is_grad_enabled()
            and not model.pack_seq and not torch.cuda.is_current_stream_capturing())


def _give_up(st, why):
    st.broken = why
    st.clear()
    warnings.warn("pika_amd.train_graph: the training step stays an eager launch sequence (%s)" % why)


def _capture(model, st, key, x, y, x_len, t_valid=None):
    from .rnnt import CompactGrad, LazyDenseGrad, LazyLogProbs
    dev = x.device
    e = _Entry()
    e.key, e.gen = key, 0
    e.inputs = [x.clone(), y.clone(), None if x_len is None else x_len.clone()]
    e.labels32 = y.to(torch.int32)       # what the loss reads when the label axis was padded to its bucket
    e.t_valid = None if t_valid is None else torch.tensor([int(t_valid)], dtype=torch.int32, device=dev)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    params = [p for _, p in named]
    # The recording differentiates fresh leaf ALIASES of the parameters (same storage), not the parameters themselves: the
    # reference loop keeps `loss` / `outputs` of the previous iteration alive across the next forward, and with them the
    # parameters' AccumulateGrad nodes, which belong to the stream that step ran on -- the autograd engine would then make
    # THAT stream wait for the capturing one inside the capture (torch warns "AccumulateGrad node's stream does not match";
    # hipStreamEndCapture crashes).  The aliases' nodes are born on the capture stream.
    aliases = {n: p.detach().requires_grad_(True) for n, p in named}
    if st.pool is None:
        st.pool = torch.cuda.graph_pool_handle()
    lib = _lib.lib()
    e.gf, e.gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    if isinstance(getattr(model, "decoder", None), torch.nn.LSTM) and y.dim() == 2:
        from .model import lstm
        lstm.reserve(model.decoder, y.shape[0], y.shape[1] + 1, dev)        # (SOS + labels: the recurrence's steps)
    # thread_local: the loader thread keeps issuing its own uploads / kernels on its side stream during the capture
    # (not torch.func.functional_call: with a module registered under two names -- the embedding the prediction network
    # shares with the transducer -- it leaves the alias in place of the Parameter when it restores the module)
    by_id = {id(p): aliases[n] for n, p in named}
    swapped = []
    for mod in model.modules():
        for k, p in list(mod._parameters.items()):
            if p is not None and id(p) in by_id:
                swapped.append((mod, k, p))
                mod._parameters[k] = by_id[id(p)]
    try:
        with torch.cuda.graph(e.gf, pool=st.pool, capture_error_mode="thread_local"):
            out = model._forward_eager(e.inputs[0], e.inputs[1], e.inputs[2], True, valid_frames=e.t_valid)
    finally:
        for mod, k, p in swapped:
            mod._parameters[k] = p
    compact_ok = (isinstance(out, LazyLogProbs) and getattr(out, "_pika_lazy_grad_ok", False) and out.state.scale == 1.0
                  and out.dim() == 4)
    if compact_ok:
        # the fast form: raw logits out, the loss' compact gradient in (neither log-probs nor a dense gradient exist)
        e.kind = "compact"
        B, T, U1, V = out.buf.shape          # (the kernels' columns: more than the caller's when the joint padded its layer)
        e.logits, e.partials, e.scale = out.buf, out.state.partials, out.state.scale
        e.gathered, e.recompute = out.state.gathered, out.state.recompute      # 16-bit logits (JointOutFn.forward)
        e.dims = (B, T, U1, V, 0)
        with torch.cuda.device(dev):
            e.ws = torch.empty(int(lib.pika_rnnt_workspace_bytes(B, T, U1)), dtype=torch.uint8, device=dev)
            e.lse = torch.empty(B * T * U1, dtype=torch.float32, device=dev)
        compact = CompactGrad.__new__(CompactGrad)
        compact.ws, compact.dims, compact.ptr, compact.version = e.ws, e.dims, 0, 0
        gout = LazyDenseGrad(compact, None, None, None, width=None if out.shape[-1] == V else out.shape[-1])
        gout.lse = e.lse
    elif type(out) is torch.Tensor and out.requires_grad:
        # the general form (vocabularies the lazy joint does not take, PIKA_LAZY_LOGPROBS=0, ...): a plain output tensor
        # and a dense gradient copied into a static buffer, as torch.cuda.make_graphed_callables does it
        e.kind = "dense"
        e.logits, e.partials, e.scale, e.ws, e.lse, e.dims = out, None, 1.0, None, None, None
        e.gathered = e.recompute = None
        with torch.cuda.device(dev):
            gout = e.gout = torch.empty_like(out)
    else:
        del out
        return None, "the model's output is neither this package's lazy log-probs nor a plain tensor"
    err = None
    with torch.cuda.graph(e.gb, pool=st.pool, capture_error_mode="thread_local"):
        try:
            grads = torch.autograd.grad((out,), [aliases[n] for n, _ in named], grad_outputs=(gout,), allow_unused=True)
            # what AccumulateGrad would make of them: dense tensors of the parameter's dtype and layout (a transposed or
            # expanded gradient would send clip / SGD to the stock multi-tensor path); the copies are part of graph B
            grads = [g if g is None or (g.dtype == p.dtype and g.is_contiguous() and g.shape == p.shape)
                     else g.to(p.dtype).expand_as(p).contiguous() for g, (_, p) in zip(grads, named)]
            grads = distinct_buffers(grads)
        except Exception as ex:         # reported below; leaving the context with an exception in flight ends the capture twice
            err = ex
            if os.environ.get("PIKA_TRAIN_GRAPH_DEBUG"):
                import traceback
                traceback.print_exc()
    if err is not None:
        del out
        return None, "backward capture: %s: %s" % (type(err).__name__, str(err).split("\n")[0])
    del out
    if os.environ.get("PIKA_TRAIN_GRAPH_DEBUG"):
        print("train_graph capture: kind %s, no gradient for %s" % (e.kind, [n for (n, _), g in zip(named, grads) if g is None]),
              flush=True)
    e.grads = [(p, g) for p, g in zip(params, grads) if g is not None]
    return e, None


def _memory_to_spare(device):
    try:
        free, total = torch.cuda.mem_get_info(device)
    except Exception:
        return True
    return free > 0.25 * total


def distinct_buffers(grads):
    """autograd.grad returns ONE tensor for two parameters whose sum entered the model (`b_ih + b_hh`: the addition's backward
    hands its incoming gradient to both); AccumulateGrad would give every parameter a buffer of its own -- the clip and
    the optimizer scale gradients in place, a shared buffer would be scaled twice."""
    seen, out = set(), []
    for g in grads:
        if g is not None:
            if g.data_ptr() in seen:
                g = g.clone()
            seen.add(g.data_ptr())
        out.append(g)
    return out


class _GraphedFn(torch.autograd.Function):
    """The autograd node of a replayed forward: `forward` wraps the static logits, `backward` replays graph B."""

    @staticmethod
    def forward(ctx, st, e, *params):
        from .rnnt import LazyLogProbs, LogitsState
        ctx.set_materialize_grads(False)
        if e.kind == "dense":
            ctx.st, ctx.e, ctx.gen, ctx.state, ctx.n = st, e, e.gen, None, len(params)
            return e.logits.detach()
        state = LogitsState(e.scale)
        state.partials, state.gathered, state.recompute = e.partials, e.gathered, e.recompute
        ctx.st, ctx.e, ctx.gen, ctx.state, ctx.n = st, e, e.gen, state, len(params)
        return LazyLogProbs(state, e.logits)

    @staticmethod
    def backward(ctx, g):
        from .rnnt import LazyDenseGrad
        st, e = ctx.st, ctx.e
        none = (None,) * (2 + ctx.n)
        if g is None:
            return none
        if st.last is None or st.last[0] is not e or st.last[1] != ctx.gen or e.gen != ctx.gen:
            raise RuntimeError("pika_amd.train_graph: backward of a forward that is not the model's latest one (its static "
                               "buffers have been overwritten); PIKA_TRAIN_GRAPH=0 runs such a loop eagerly")
        if e.kind == "dense":
            e.gout.copy_(g.dense() if isinstance(g, LazyDenseGrad) else g)
        elif not ctx.state.raw:
            raise RuntimeError("pika_amd.train_graph: the log-probs were read between forward and backward; the captured "
                               "backward differentiates the raw logits this package's loss reads (and a read normalises an "
                               "fp32 logits buffer in place); PIKA_TRAIN_GRAPH=0 runs such a loop eagerly")
        elif not (isinstance(g, LazyDenseGrad) and g._dense is None and g.lse is not None
                  and tuple(g.compact.dims) == tuple(e.dims)):
            raise RuntimeError("pika_amd.train_graph: the gradient of the model's output is not the compact gradient of this "
                               "package's RNNTLoss(blank=0) on the raw logits (got %r); PIKA_TRAIN_GRAPH=0 runs such a loop "
                               "eagerly" % (g,))
        else:
            if g.compact.ws.data_ptr() != e.ws.data_ptr():
                e.ws.copy_(g.compact.ws)
            if g.lse.data_ptr() != e.lse.data_ptr():
                e.lse.copy_(g.lse)
        # a p.grad that IS the static tensor (the caller did not zero its gradients, or backs through a retained graph
        # twice) holds the earlier values the replay is about to overwrite: accumulate as autograd would
        kept = {id(gr): gr.clone() for p, gr in e.grads if p.grad is gr}
        dbg_sync = int(DEFAULTS["debug_sync"])
        if dbg_sync & 4:
            torch.cuda.synchronize()
        e.gb.replay()
        if dbg_sync & 8:
            torch.cuda.synchronize()
        if os.environ.get("PIKA_TRAIN_GRAPH_DEBUG") == "3":     # diagnostics: where does a non-finite value first appear?
            names = {id(p): n for n, p in st.model_ref().named_parameters()} if st.model_ref() is not None else {}
            bad = [(names.get(id(p), "?"), float(gr.abs().max())) for p, gr in e.grads if not bool(torch.isfinite(gr).all())]
            big = max(float(gr.abs().max()) for _, gr in e.grads)
            print("train_graph call %d: after backward replay: max |grad| %.3e, non-finite in %d tensors %s" % (
                st.calls, big, len(bad), bad[:6]), flush=True)
        for p, gr in e.grads:
            if p.grad is None:
                p.grad = gr
            elif p.grad is gr:
                p.grad = kept[id(gr)] + gr
            else:
                p.grad.add_(gr)
        return none


def forward(model, x, y, x_len, softmax):
    """`Net.forward` when `wanted(...)`: eager during warm-up / for shapes not (yet) captured, otherwise a replay."""
    st = model._step_graphs
    st.calls += 1
    if not st.freeze_salt:
        st.salt.random_()                                # device-side draw: new dropout masks, no host involvement
    if st.calls <= st.warmup:
        st.stats["eager"] += 1
        return model._forward_eager(x, y, x_len, softmax)
    # the parameter list is walked once (300 tensors through 1200 modules: 0.7 ms of Python per step); every call checks
    # that the list still is the model's (count, first / middle / last object) and that nothing re-pointed the storage
    # (BMUF flattens the parameters into its vector; .to(); load_state_dict(assign=True)): then the graphs are dropped
    params = st.params
    if params is None or (st.calls & 63) == 0:
        params = st.params = [p for p in model.parameters() if p.requires_grad]
    ptrs = tuple(p.data_ptr() for p in params)
    if st.param_ptrs != ptrs:
        st.clear()
        st.param_ptrs = ptrs
    # Shapes.  The LABEL axis may be padded with the embedding's padding index -- what the loader itself pads the shorter
    # utterances of a batch with (otf_utt_loader.py:262-270): the prediction network masks those positions as keys and is
    # causal, the loss never reads lattice columns beyond an utterance's label count.  The TIME axis may be padded when
    # the model's encoder takes `valid_frames` (st.t_bucket > 0): BatchNorm statistics / counts / gradients then run over
    # the data frames only and padding frames are no attention keys (model/encoder.py), so the values of every lattice cell
    # the loss reads and every parameter gradient are those of the unpadded batch (tests/test_train_step_gpu.py).
    # A batch is served by an existing pair of graphs whose label axis is up to u_bucket - 1 labels and whose time axis is
    # up to t_bucket - 1 frames longer than its own; a NEW pair is captured at the batch's own shape, so a corpus (or a
    # benchmark) of one shape pays for no padding.
    T, U = x.shape[1], y.shape[1]
    pad = getattr(model.embed, "padding_idx", None)
    timed = st.t_bucket > 0 and x.dim() == 3 and hasattr(model.encoder, "hidden_conv")     # the TDNN-Transformer encoder

    def key_for(frames, width):
        return ((x.shape[0], frames) + tuple(x.shape[2:]), x.dtype, (y.shape[0], width), y.dtype,
                None if x_len is None else (tuple(x_len.shape), x_len.dtype), timed)
    e, Tb, Ub = None, T, U
    widths = range(U, U + (st.u_bucket if (pad is not None and y.dim() == 2) else 1))
    for k_, cand in st.entries.items():      # (at most max_graphs entries)
        if k_[5] == timed and k_[0][0] == x.shape[0] and k_[0][2:] == tuple(x.shape[2:]) and k_[2][1] in widths \
                and T <= k_[0][1] < T + max(st.t_bucket, 1) and k_ == key_for(k_[0][1], k_[2][1]):
            if e is None or (k_[0][1], k_[2][1]) < (Tb, Ub):
                e, Tb, Ub = cand, k_[0][1], k_[2][1]
    key = key_for(Tb, Ub)
    if e is None:
        n = st.seen[key] = st.seen.get(key, 0) + 1
        if os.environ.get("PIKA_TRAIN_GRAPH_DEBUG") == "2":
            print("train_graph call %d: %s seen %d" % (st.calls, key[0:3:2], n), flush=True)
        cx, cy = x, y
        if n < st.min_seen:
            # A corpus whose lengths vary from batch to batch shows an exact shape once: its batches fall into BUCKETS of
            # t_bucket frames x u_bucket labels instead.  A bucket that has shown two different shapes (and min_seen
            # batches) gets a pair of graphs at its upper boundary, which every later batch of the bucket rides padded --
            # while there is room (more live buckets than max_graphs: the rest stay the eager launch sequence).
            bT = -(-T // st.t_bucket) * st.t_bucket if timed else T
            bU = -(-U // st.u_bucket) * st.u_bucket if (pad is not None and y.dim() == 2) else U
            shapes = st.bucket_shapes.setdefault(key_for(bT, bU), set())
            shapes.add((T, U))
            # (a bucket never pushes out another pair: with more live buckets than max_graphs the table would turn over for
            #  ever -- tools/real_corpus_sim.py: a capture every ~12 steps at 4 pairs and ~20 buckets -- and never while less
            #  than a quarter of the device memory is free: a pair holds its static logits and gradients)
            room = len(st.entries) < st.max_graphs and _memory_to_spare(x.device)
            if not (DEFAULTS["bucket_capture"] and (bT, bU) != (T, U) and len(shapes) >= max(2, st.min_seen) and room):
                st.stats["eager"] += 1
                return model._forward_eager(x, y, x_len, softmax)
            key, Tb, Ub = key_for(bT, bU), bT, bU
            cx = x.new_zeros((x.shape[0], bT) + tuple(x.shape[2:]))
            cx[:, :T] = x
            cy = y.new_full((y.shape[0], bU), int(pad)) if bU != U else y
            if bU != U:
                cy[:, :U] = y
            st.stats["bucket_captures"] = st.stats.get("bucket_captures", 0) + 1
        while len(st.entries) >= st.max_graphs:
            old_key, _ = st.entries.popitem(last=False)  # least recently used: its static outputs go back to the pool
            st.used_at.pop(old_key, None)
            st.stats["evictions"] += 1
        try:
            e, why = _capture(model, st, key, cx, cy, x_len, t_valid=T if timed else None)
        except Exception as err:                         # a launch the stream capture refuses, out of memory, ...
            e, why = None, "%s: %s" % (type(err).__name__, str(err).split("\n")[0])
        if e is None:
            _give_up(st, why)
            st.stats["eager"] += 1
            return model._forward_eager(x, y, x_len, softmax)
        st.entries[key] = e
        st.stats["captures"] += 1
    else:
        st.entries.move_to_end(key)
        if Tb != T:
            e.inputs[0][:, :T].copy_(x, non_blocking=True)      # frames beyond T keep what they held: finite, and masked
            st.stats["padded"] = st.stats.get("padded", 0) + 1
        else:
            e.inputs[0].copy_(x, non_blocking=True)
        if e.t_valid is not None:
            e.t_valid.fill_(T)
        if Ub != U:
            e.inputs[1].fill_(int(pad))
            e.inputs[1][:, :U].copy_(y, non_blocking=True)
        else:
            e.inputs[1].copy_(y, non_blocking=True)
        if x_len is not None:
            e.inputs[2].copy_(x_len, non_blocking=True)
        e.labels32.copy_(e.inputs[1], non_blocking=True)
    if os.environ.get("PIKA_TRAIN_GRAPH_DEBUG") == "2":
        print("train_graph call %d: replay of %s for labels %s" % (st.calls, key[0:3:2], tuple(y.shape)), flush=True)
    e.gen += 1
    st.last = (e, e.gen)
    st.used_at[key] = st.calls
    dbg_sync = int(DEFAULTS["debug_sync"])      # debugging: 1 before F, 2 after F, 4 before B, 8 after B
    if dbg_sync & 1:
        torch.cuda.synchronize()
    e.gf.replay()
    if dbg_sync & 2:
        torch.cuda.synchronize()
    st.stats["replays"] += 1
    if os.environ.get("PIKA_TRAIN_GRAPH_DEBUG") == "3":
        print("train_graph call %d: after forward replay: logits finite %s, max |logit| %.3e, input max %.3e" % (
            st.calls, bool(torch.isfinite(e.logits).all()), float(e.logits.abs().max()), float(e.inputs[0].abs().max())),
            flush=True)
    out = _GraphedFn.apply(st, e, *params)
    if e.kind == "compact":
        out._pika_lazy_grad_ok = True
        out._pika_loss_buffers = (e.ws, e.lse)           # the loss writes its workspace / lse where graph B reads them
    out._pika_labels = e.labels32                        # the labels over the padded label axis
    return out


class GraphedTrainStep(object):
    """The reference loop's step, spelled out, on a model with the graphs enabled (bench.py, tests): zero_grad,
    forward, loss, backward, inf-norm clip, optimizer step.  `optimizer` may be replaced / re-parameterised between
    calls (the script rebuilds it after every BMUF block, :115-123): nothing of it is captured."""

    def __init__(self, model, loss_fn, make_optimizer, clip=3.0, warmup=2, max_graphs=None, min_seen=None, t_bucket=0):
        self.model, self.loss_fn, self.clip = model, loss_fn, float(clip)
        self.make_optimizer = make_optimizer
        self.optimizer = make_optimizer()
        disable(model)
        self.state = enable(model, warmup=warmup, max_graphs=max_graphs, min_seen=1 if min_seen is None else min_seen,
                            t_bucket=t_bucket)

    @property
    def graphs(self):
        return self.state.entries

    @property
    def salt(self):
        return self.state.salt

    def close(self):
        disable(self.model)

    def reset_momentum(self):
        """What re-creating the optimizer does to its state (train_transducer_bmuf_otfaug.py:121)."""
        self.optimizer = self.make_optimizer()

    def set_lr(self, lr):
        for g in self.optimizer.param_groups:
            g["lr"] = float(lr)

    def __call__(self, data, labels, len_b, ali):
        """data (B,T,F) f32 after CMVN / SpecAugment, labels (B,U), len_b / ali (B,) int32: one optimisation step;
        returns the summed loss as a device tensor."""
        self.optimizer.zero_grad(set_to_none=True)
        out = self.model(data, labels.long(), len_b, True)
        loss = self.loss_fn(out, labels.int(), len_b, ali).sum()
        loss.backward()
        if self.clip > 0:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip, norm_type=float("inf"))
        self.optimizer.step()
        return loss.detach()
