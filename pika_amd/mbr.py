"""Minimum-Bayes-risk training step, MI355X formulation of
trainer/train_transducer_mbr_bmuf_otfaug.py:112-235.

The reference, per batch: N-best decode -> encoder forward -> RNN-T loss backward -> softmax over
the N-best scores, edit distances, `seq_grad = prob * (dist - E[dist])` (:171-195) -> prediction net
on the N-best label sequences -> for every hypothesis walk its (t,u) trajectory, gather encoder /
prediction vectors into a (B*beam, T+U, 2H) tensor, joint + log-softmax, and back-propagate a DENSE
(B*beam, T+U, V) gradient that holds one non-zero per row (blank entries scaled by 1/T) (:197-235).

Here: trajectories come from two cumulative sums on the device; the joint uses the split fc1/fc_gate
halves (encoder half computed once per utterance, not per hypothesis); the dense one-hot gradient
never exists -- `RiskFn` returns the surrogate  sum_rows val * log_softmax(scale*logits)[row, sym]
and its backward writes d/dlogits in place with one HIP kernel (pika_mbr_risk_grad_rows).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .model import ops


def _ints(h):
    """A hypothesis as a list of ints: the decoder hands back lists of 0-dim tensors (what the reference's scripts
    index), one conversion per hypothesis instead of one per symbol."""
    if len(h) and torch.is_tensor(h[0]):
        return torch.stack(list(h)).tolist()
    return [int(e) for e in h]


def edit_distances(pairs):
    """Levenshtein distance of every (a, b) pair of int sequences in ONE library call (pika_edit_distances: host
    code of libpika_amd.so; the reference calls editdistance.eval per hypothesis)."""
    if not pairs:
        return []
    flat, a_off, a_len, b_off, b_len = [], [], [], [], []
    for a, b in pairs:
        a_off.append(len(flat)); a_len.append(len(a)); flat.extend(a)
        b_off.append(len(flat)); b_len.append(len(b)); flat.extend(b)
    seqs = np.asarray(flat if flat else [0], dtype=np.int32)
    ao, bo = np.asarray(a_off, dtype=np.int64), np.asarray(b_off, dtype=np.int64)
    al, bl = np.asarray(a_len, dtype=np.int32), np.asarray(b_len, dtype=np.int32)
    out = np.zeros(len(pairs), dtype=np.int32)
    _lib.check(_lib.lib().pika_edit_distances(seqs.ctypes.data, ao.ctypes.data, al.ctypes.data, bo.ctypes.data,
                                              bl.ctypes.data, len(pairs), out.ctypes.data), "pika_edit_distances")
    return out.tolist()


def edit_distance(a, b):
    a, b = list(a), list(b)
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


class RiskFn(torch.autograd.Function):
    """surrogate = sum_r val[r] * log_softmax(scale * logits)[r, sym[r]]  (logits overwritten)."""

    @staticmethod
    def forward(ctx, logits, sym, val, scale):
        rows, V = logits.shape
        if logits.is_cuda:
            with torch.cuda.device(logits.device):
                _lib.check(_lib.lib().pika_log_softmax_rows(logits.data_ptr(), rows, V, V, float(scale),
                                                            torch.cuda.current_stream().cuda_stream),
                           "pika_log_softmax_rows")
            lp = logits  # overwritten in place by the kernel; its producer (fc2 GEMM) keeps no copy
        else:
            lp = F.log_softmax(scale * logits, dim=-1)
        ctx.scale = float(scale)
        ctx.save_for_backward(lp, sym, val)
        return (lp.gather(1, sym.long().unsqueeze(1)).squeeze(1) * val).sum()

    @staticmethod
    def backward(ctx, g):
        lp, sym, val = ctx.saved_tensors
        rows, V = lp.shape
        v = (val * g).float().contiguous()
        if lp.is_cuda:
            with torch.cuda.device(lp.device):
                _lib.check(_lib.lib().pika_mbr_risk_grad_rows(
                    lp.data_ptr(), sym.data_ptr(), v.data_ptr(), rows, V, V, ctx.scale,
                    torch.cuda.current_stream().cuda_stream), "pika_mbr_risk_grad_rows")
            return lp, None, None, None
        onehot = F.one_hot(sym.long(), V).to(lp.dtype)
        return ctx.scale * v.unsqueeze(1) * (onehot - lp.exp()), None, None, None


def risk_terms(hyps, scores, targets, target_lens, blk, device):
    """:171-195.  hyps[b][j]: symbol sequence incl. blanks; scores[b][j].  Returns prob, dist,
    seq_grad (B,beam) and the blank-free hypotheses."""
    B, beam = len(hyps), len(hyps[0])
    prob = F.softmax(torch.tensor([[float(s) for s in row] for row in scores], device=device), dim=1)
    nonblk = [[[e for e in _ints(h) if e != blk] for h in row] for row in hyps]
    tl = [int(v) for v in torch.as_tensor(target_lens).tolist()]
    refs = torch.as_tensor(targets).tolist()
    d = edit_distances([(refs[b][:tl[b]], nonblk[b][j]) for b in range(B) for j in range(beam)])
    dist = torch.tensor(d, dtype=torch.float32).view(B, beam).to(device)
    avg = (prob * dist).sum(dim=1, keepdim=True)
    return prob, dist, prob * (dist - avg), nonblk


def hyp_arrays(hyps, nonblk, pad, blk, S=None, Umax=None):
    """The N-best of a batch as three host arrays (ONE upload each): y (B*beam, Umax) blank-free labels padded with the
    embedding's padding index, sym (B*beam, S) symbols incl. blanks padded with blank, slen (B*beam,) symbol counts.
    S / Umax may be given larger than the batch needs (the graphed step pads them to its buckets)."""
    B, beam = len(hyps), len(hyps[0])
    u_need = max(len(h) for row in nonblk for h in row)
    s_need = max(max(len(h) for row in hyps for h in row), 1)
    Umax = u_need if Umax is None else Umax
    S = s_need if S is None else S
    assert Umax >= u_need and S >= s_need
    y_h = np.full((B * beam, max(Umax, 0)), pad, dtype=np.int64)
    sym_h = np.full((B * beam, S), blk, dtype=np.int64)
    slen_h = np.zeros(B * beam, dtype=np.int64)
    for b in range(B):
        for j in range(beam):
            r = b * beam + j
            if nonblk[b][j]:
                y_h[r, :len(nonblk[b][j])] = nonblk[b][j]
            h = _ints(hyps[b][j])
            if h:
                sym_h[r, :len(h)] = h
            slen_h[r] = len(h)
    return y_h, sym_h, slen_h


def risk_surrogate(model, enc, y, sym, slen, seq_grad, blk, sm_scale):
    """:197-235 as a differentiable scalar: sum over the live trajectory rows of val * log_softmax(sm_scale * logits)[sym]
    with val = seq_grad of the row's hypothesis (blank rows / T).  Its gradient IS what `out.backward(mbr_grad)` of the
    reference back-propagates.  Device tensors in, no host reads: capturable (GraphedMbrStep)."""
    bb, S = sym.shape
    B = enc.shape[0]
    beam = bb // B
    dev = enc.device
    T, H = enc.shape[1], enc.shape[2]
    sos = torch.zeros(bb, 1, dtype=torch.long, device=dev)
    pred = model.predict(torch.cat((sos, y), dim=1))                          # (bb, U, H)   :198-206
    is_blk = sym.eq(blk)
    steps = torch.arange(S, device=dev).unsqueeze(0)
    live = steps < slen.unsqueeze(1)
    t_idx = (torch.cumsum(is_blk & live, 1) - (is_blk & live).long()).clamp(max=T - 1)
    u_idx = torch.cumsum(~is_blk & live, 1) - (~is_blk & live).long()
    rows_b = torch.arange(B, device=dev).repeat_interleave(beam).unsqueeze(1).expand(-1, S)
    rows_r = torch.arange(bb, device=dev).unsqueeze(1).expand(-1, S)
    w1, wg = model.fc1, model.fc_gate
    e1 = ops.linear(enc, w1.weight[:, :H].contiguous(), w1.bias)
    eg = ops.linear(enc, wg.weight[:, :H].contiguous(), wg.bias)
    p1 = ops.linear(pred, w1.weight[:, H:].contiguous())
    pg = ops.linear(pred, wg.weight[:, H:].contiguous())
    z1 = e1[rows_b, t_idx] + p1[rows_r, u_idx]
    zg = eg[rows_b, t_idx] + pg[rows_r, u_idx]
    h = torch.tanh(z1) * torch.sigmoid(zg)
    logits = ops.linear(h.reshape(-1, H), model.fc2.weight, model.fc2.bias)
    val = seq_grad.reshape(-1, 1).expand(-1, S) * live
    val = torch.where(is_blk, val / float(T), val)
    return RiskFn.apply(logits, sym.reshape(-1).int(), val.reshape(-1).float().contiguous(), sm_scale)


def mbr_backward(model, enc, hyps, seq_grad, nonblk, blk, sm_scale):
    """:197-235 -- accumulates the risk gradient into the model's .grad through `enc` (B,T,H, part
    of the live graph) and the prediction net.  Returns the surrogate value (for tests)."""
    dev = enc.device
    y_h, sym_h, slen_h = hyp_arrays(hyps, nonblk, model.embed.padding_idx, blk)
    y, sym, slen = (torch.from_numpy(a).to(dev) for a in (y_h, sym_h, slen_h))
    # trajectory: before step s the path has consumed t = #blanks, u = #labels of steps < s  (:212-217); one non-zero per
    # live row: seq_grad at the emitted symbol, blank entries scaled by 1/T (:225-233) -- risk_surrogate
    surrogate = risk_surrogate(model, enc, y, sym, slen, seq_grad, blk, sm_scale)
    surrogate.backward()
    return float(surrogate.detach())


class _MbrEntry(object):
    __slots__ = ("graph", "x", "labels", "labels32", "x_len", "ali", "y", "sym", "slen", "seq_grad", "rnnt", "grads", "key", "t_valid")


class GraphedMbrStep(object):
    """The TRAINING half of the MBR step (train_transducer_mbr_bmuf_otfaug.py:120-235: encoder forward, RNN-T loss on the
    reference labels, risk terms, prediction net on the N-best, trajectory joint, both backward passes) as ONE hipGraph per
    batch shape.  The script's loop calls model.encoder / .decoder / .fc1 ... inline and back-propagates twice through the
    encoder (`rnnt_loss.backward(retain_graph=True)`, then `out.backward(mbr_grad)`); here both losses hang off one recorded
    forward and ONE backward pass differentiates `rnnt_scale * rnnt + surrogate` -- the same gradients (the encoder's
    backward sees the sum of the two d(enc) instead of accumulating two passes), ~half the encoder backward work, and no
    launch is issued from Python in the steady state: eager, the ~1300 launches of this half take 46 ms of host time for
    ~14 ms of device work at B = 8.

        step = GraphedMbrStep(model, rnnt_scale=0.1, sm_scale=0.8, blk=0)
        rnnt = step(feats, labels, x_len, ali, hyps, scores)      # p.grad of every parameter holds the step's gradient
        clip_grad_norm_(...); optimizer.step()                    # eager, three launches (pika_amd/optim.py)

    Shapes: a graph is keyed by (feats shape, label-axis width, beam, S bucket, U bucket) -- S = the longest hypothesis
    incl. blanks, padded up to a multiple of `s_bucket` with dead rows (val = 0), U = the longest blank-free hypothesis,
    padded to a multiple of `u_bucket` with the embedding's padding index (masked as keys by the prediction network).  A key
    is captured the `min_seen`-th time it appears (LRU bound `max_graphs`, ONE memory pool); until then, and whenever a
    capture fails, the step runs as the eager launch sequence (`eager_step`: the two backward passes of the script).
    A batch also rides a graph whose time axis is up to `t_bucket` - 1 frames and whose label axis is up to `l_bucket` - 1
    labels longer than its own (padding frames masked through the encoder's `valid_frames`, labels padded with the padding
    index, as pika_amd.train_graph does for the plain step), and a bucket of t_bucket x l_bucket that has shown two different
    shapes is captured at its upper boundary: a corpus whose batch shapes never recur still gets graphs.
    Dropout: the device-side salt word of pika_amd.train_graph, re-drawn before every replay."""

    def __init__(self, model, rnnt_scale=1.0, sm_scale=1.0, blk=0, max_graphs=16, min_seen=2, s_bucket=32, u_bucket=8,
                 warmup=1, t_bucket=64, l_bucket=8):
        from . import train_graph
        from .rnnt import RNNTLoss
        self.model, self.rnnt_scale, self.sm_scale, self.blk = model, float(rnnt_scale), float(sm_scale), int(blk)
        self.max_graphs, self.min_seen = max(1, int(max_graphs)), max(1, int(min_seen))
        # (at least ONE eager call first: libraries that initialise lazily -- hipBLASLt behind a stock torch op of a narrow layer,
        #  kernel attributes, the allocator -- must not meet their first call inside a stream capture: hipBLASLt aborts the process)
        self.s_bucket, self.u_bucket, self.warmup = max(1, int(s_bucket)), max(1, int(u_bucket)), max(1, int(warmup))
        self.loss = RNNTLoss(blank=self.blk).apply
        import collections
        self.entries = collections.OrderedDict()
        self.seen, self.calls, self.pool, self.broken = {}, 0, None, None
        self.t_bucket, self.l_bucket = max(0, int(t_bucket)), max(1, int(l_bucket))
        self.bucket_shapes, self.used_at = {}, {}
        self.param_ptrs = None
        self.stats = {"replays": 0, "captures": 0, "eager": 0, "evictions": 0}
        self.salt = train_graph._salt_acquire(next(model.parameters()).device)
        self._closed = False
        self.last_risk = None        # (prob, dist) of the latest call, device tensors: expected risk = (prob * dist).sum()

    def close(self):
        if not self._closed:
            from . import train_graph
            self._closed = True
            self.entries.clear()
            train_graph._salt_release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the step as the script runs it (two backward passes), also the fallback -------------------------------------------
    def eager_step(self, feats, labels, x_len, ali, hyps, scores, terms=None):
        model = self.model
        B = feats.shape[0]
        enc = model.encode(feats, None)                                       # :124-138
        sos = torch.zeros(B, 1, dtype=torch.long, device=feats.device)
        pred = model.predict(torch.cat((sos, labels.long()), dim=1))
        lp = ops.joint(enc, pred, model.fc1, model.fc_gate, model.fc2, log_softmax=True)
        rnnt = self.rnnt_scale * self.loss(lp, labels.int(), x_len.int(), ali.int()).sum()      # :152-158
        rnnt.backward(retain_graph=True)
        prob, dist, seq_grad, nonblk = terms if terms is not None else risk_terms(hyps, scores, labels, ali, self.blk,
                                                                                  feats.device)
        self.last_risk = (prob, dist)
        mbr_backward(model, enc, hyps, seq_grad, nonblk, self.blk, self.sm_scale)                # :197-235
        return rnnt.detach()

    def _capture(self, key, feats, labels, x_len, ali, y, sym, slen, seq_grad, t_valid=None):
        from . import train_graph
        model = self.model
        dev = feats.device
        e = _MbrEntry()
        e.key = key
        e.x, e.labels, e.x_len, e.ali = feats.clone(), labels.long().clone(), x_len.int().clone(), ali.int().clone()
        # (frames of data on a time axis that may be padded: a device word the encoder's BatchNorm / attention launches read)
        e.t_valid = None if t_valid is None else torch.tensor([int(t_valid)], dtype=torch.int32, device=dev)
        e.labels32 = e.labels.int()
        e.y, e.sym, e.slen, e.seq_grad = y.clone(), sym.clone(), slen.clone(), seq_grad.float().clone()
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        # fresh leaf aliases of the parameters, for the reason pika_amd/train_graph.py::_capture gives (AccumulateGrad
        # nodes of an earlier eager step belong to another stream)
        aliases = {n: p.detach().requires_grad_(True) for n, p in named}
        by_id = {id(p): aliases[n] for n, p in named}
        swapped = []
        for mod in model.modules():
            for k, p in list(mod._parameters.items()):
                if p is not None and id(p) in by_id:
                    swapped.append((mod, k, p))
                    mod._parameters[k] = by_id[id(p)]
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        if isinstance(getattr(model, "decoder", None), torch.nn.LSTM):
            from .model import lstm
            # (the recurrence runs over SOS + labels for the RNN-T part and over the N-best rows for the risk part)
            lstm.reserve(model.decoder, max(e.labels.shape[0], e.y.shape[0]), max(e.labels.shape[1], e.y.shape[1]) + 1, dev)
        e.graph = torch.cuda.CUDAGraph()
        err = None
        try:
            with torch.cuda.graph(e.graph, pool=self.pool, capture_error_mode="thread_local"):
                try:
                    B = e.x.shape[0]
                    enc = model.encode(e.x, None) if e.t_valid is None else model.encode(e.x, None, valid_frames=e.t_valid)
                    sos = torch.zeros(B, 1, dtype=torch.long, device=dev)
                    pred = model.predict(torch.cat((sos, e.labels), dim=1))
                    lp = ops.joint(enc, pred, model.fc1, model.fc_gate, model.fc2, log_softmax=True, labels=e.labels)
                    rnnt = self.rnnt_scale * self.loss(lp, e.labels32, e.x_len, e.ali).sum()
                    surrogate = risk_surrogate(model, enc, e.y, e.sym, e.slen, e.seq_grad, self.blk, self.sm_scale)
                    grads = torch.autograd.grad((rnnt + surrogate,), [aliases[n] for n, _ in named], allow_unused=True)
                    grads = [g if g is None or (g.dtype == p.dtype and g.is_contiguous() and g.shape == p.shape)
                             else g.to(p.dtype).expand_as(p).contiguous() for g, (_, p) in zip(grads, named)]
                    grads = train_graph.distinct_buffers(grads)
                    e.rnnt = rnnt.detach()
                except Exception as ex:     # leaving the context with an exception in flight ends the capture twice
                    err = ex
        finally:
            for mod, k, p in swapped:
                mod._parameters[k] = p
        if err is not None:
            return None, "%s: %s" % (type(err).__name__, str(err).split("\n")[0])
        e.grads = [(p, g) for (_, p), g in zip(named, grads) if g is not None]
        return e, None

    def __call__(self, feats, labels, x_len, ali, hyps, scores):
        """feats (B,T,F) f32, labels (B,U) with the padding index beyond ali, x_len / ali (B,), hyps / scores = the N-best of
        TransducerDecoder.decode_batch.  Leaves the step's gradient in p.grad (accumulating into gradients the caller left
        in place) and returns rnnt_scale * sum of the RNN-T costs (device tensor)."""
        import os
        import warnings
        from . import train_graph
        model = self.model
        dev = feats.device
        self.calls += 1
        terms = risk_terms(hyps, scores, labels, ali, self.blk, dev)          # :163-195
        prob, dist, seq_grad, nonblk = terms
        self.last_risk = (prob, dist)
        if (self.broken is not None or self.calls <= self.warmup or not feats.is_cuda or self._closed
                or os.environ.get("PIKA_TRAIN_GRAPH", "1") == "0" or getattr(model, "pack_seq", False)
                or torch.cuda.is_current_stream_capturing()):
            self.stats["eager"] += 1
            return self.eager_step(feats, labels, x_len, ali, hyps, scores, terms)
        import pika_amd
        if not pika_amd.HIP_GRAPHS_SAFE_TO_ALTERNATE:
            self.broken = "HIP graph packet capture is on (pika_amd/__init__.py)"
            self.stats["eager"] += 1
            return self.eager_step(feats, labels, x_len, ali, hyps, scores, terms)
        ptrs = tuple(p.data_ptr() for p in model.parameters() if p.requires_grad)
        if self.param_ptrs != ptrs:           # BMUF re-pointed the parameters into its flat vector, .to(), ...
            self.entries.clear()
            self.param_ptrs = ptrs
        B, beam = len(hyps), len(hyps[0])
        u_need = max(len(h) for row in nonblk for h in row)
        s_need = max(max(len(h) for row in hyps for h in row), 1)
        Sb = -(-s_need // self.s_bucket) * self.s_bucket
        Ub = -(-max(u_need, 1) // self.u_bucket) * self.u_bucket
        pad = model.embed.padding_idx
        y_h, sym_h, slen_h = hyp_arrays(hyps, nonblk, pad, self.blk, S=Sb, Umax=Ub)
        T, U = feats.shape[1], labels.shape[1]
        timed = self.t_bucket > 0 and feats.dim() == 3 and hasattr(model.encoder, "hidden_conv") and pad is not None

        def key_for(frames, width):
            return ((feats.shape[0], frames) + tuple(feats.shape[2:]), feats.dtype, (labels.shape[0], width), beam, Sb, Ub)
        e, Tb, Lb = None, T, U
        for k_, cand in self.entries.items():           # (at most max_graphs entries)
            if k_ == key_for(k_[0][1], k_[2][1]) and U <= k_[2][1] < U + (self.l_bucket if timed else 1) \
                    and T <= k_[0][1] < T + (self.t_bucket if timed else 1) and (e is None or (k_[0][1], k_[2][1]) < (Tb, Lb)):
                e, Tb, Lb = cand, k_[0][1], k_[2][1]
        key = key_for(Tb, Lb)
        if e is None:
            n = self.seen[key] = self.seen.get(key, 0) + 1
            cf, cl = feats, labels
            if n < self.min_seen:
                # (the bucket logic of pika_amd.train_graph.forward: two different shapes in a bucket -> a graph at its upper
                #  boundary while there is room: a bucket never pushes another graph out)
                bT = -(-T // self.t_bucket) * self.t_bucket if timed else T
                bL = -(-U // self.l_bucket) * self.l_bucket if timed else U
                shapes = self.bucket_shapes.setdefault(key_for(bT, bL), set())
                shapes.add((T, U))
                room = len(self.entries) < self.max_graphs and train_graph._memory_to_spare(dev)
                if not ((bT, bL) != (T, U) and len(shapes) >= max(2, self.min_seen) and room):
                    self.stats["eager"] += 1
                    return self.eager_step(feats, labels, x_len, ali, hyps, scores, terms)
                key, Tb, Lb = key_for(bT, bL), bT, bL
                cf = feats.new_zeros((feats.shape[0], bT) + tuple(feats.shape[2:]))
                cf[:, :T] = feats
                cl = labels.new_full((labels.shape[0], bL), int(pad))
                cl[:, :U] = labels
                self.stats["bucket_captures"] = self.stats.get("bucket_captures", 0) + 1
            while len(self.entries) >= self.max_graphs:
                old_key, _ = self.entries.popitem(last=False)
                self.used_at.pop(old_key, None)
                self.stats["evictions"] += 1
            y, sym, slen = (torch.from_numpy(a).to(dev) for a in (y_h, sym_h, slen_h))
            self.salt.random_()
            try:
                e, why = self._capture(key, cf, cl, x_len, ali, y, sym, slen, seq_grad, t_valid=T if timed else None)
            except Exception as err:
                e, why = None, "%s: %s" % (type(err).__name__, str(err).split("\n")[0])
            if e is None:
                self.broken = why
                warnings.warn("pika_amd.mbr.GraphedMbrStep: the step stays an eager launch sequence (%s)" % why)
                self.stats["eager"] += 1
                return self.eager_step(feats, labels, x_len, ali, hyps, scores, terms)
            self.entries[key] = e
            self.stats["captures"] += 1
        else:
            self.entries.move_to_end(key)
            if Tb != T:
                e.x[:, :T].copy_(feats, non_blocking=True)       # frames beyond T keep what they held: finite, and masked
                self.stats["padded"] = self.stats.get("padded", 0) + 1
            else:
                e.x.copy_(feats, non_blocking=True)
            if e.t_valid is not None:
                e.t_valid.fill_(T)
            if Lb != U:
                e.labels.fill_(int(pad))
                e.labels[:, :U].copy_(labels, non_blocking=True)
            else:
                e.labels.copy_(labels, non_blocking=True)
            e.labels32.copy_(e.labels, non_blocking=True)
            e.x_len.copy_(x_len, non_blocking=True)
            e.ali.copy_(ali, non_blocking=True)
            for dst, a in ((e.y, y_h), (e.sym, sym_h), (e.slen, slen_h)):
                dst.copy_(torch.from_numpy(a), non_blocking=False)      # (pageable source: the copy returns when staged)
            e.seq_grad.copy_(seq_grad, non_blocking=True)
        self.salt.random_()                              # device-side draw: new dropout masks per replay
        self.used_at[key] = self.calls
        kept = {id(g): g.clone() for p, g in e.grads if p.grad is g}
        e.graph.replay()
        self.stats["replays"] += 1
        for p, g in e.grads:
            if p.grad is None:
                p.grad = g
            elif p.grad is g:
                p.grad = kept[id(g)] + g
            else:
                p.grad.add_(g)
        return e.rnnt
