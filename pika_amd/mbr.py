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


def mbr_backward(model, enc, hyps, seq_grad, nonblk, blk, sm_scale):
    """:197-235 -- accumulates the risk gradient into the model's .grad through `enc` (B,T,H, part
    of the live graph) and the prediction net.  Returns the surrogate value (for tests)."""
    B, beam = len(hyps), len(hyps[0])
    dev = enc.device
    T, H = enc.shape[1], enc.shape[2]
    pad = model.embed.padding_idx
    Umax = max(len(h) for row in nonblk for h in row)
    S = max(max(len(h) for row in hyps for h in row), 1)
    y_h = np.full((B * beam, max(Umax, 0)), pad, dtype=np.int64)       # built on the host, ONE upload each
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
    y, sym, slen = (torch.from_numpy(a).to(dev) for a in (y_h, sym_h, slen_h))
    sos = torch.zeros(B * beam, 1, dtype=torch.long, device=dev)
    pred = model.predict(torch.cat((sos, y), dim=1))                          # (bb, U, H)   :198-206
    # trajectory: before step s the path has consumed t = #blanks, u = #labels of steps < s  (:212-217)
    is_blk = sym.eq(blk)
    steps = torch.arange(S, device=dev).unsqueeze(0)
    live = steps < slen.unsqueeze(1)
    t_idx = (torch.cumsum(is_blk & live, 1) - (is_blk & live).long()).clamp(max=T - 1)
    u_idx = torch.cumsum(~is_blk & live, 1) - (~is_blk & live).long()
    rows_b = torch.arange(B, device=dev).repeat_interleave(beam).unsqueeze(1).expand(-1, S)
    rows_r = torch.arange(B * beam, device=dev).unsqueeze(1).expand(-1, S)
    w1, wg = model.fc1, model.fc_gate
    e1 = ops.linear(enc, w1.weight[:, :H].contiguous(), w1.bias)
    eg = ops.linear(enc, wg.weight[:, :H].contiguous(), wg.bias)
    p1 = ops.linear(pred, w1.weight[:, H:].contiguous())
    pg = ops.linear(pred, wg.weight[:, H:].contiguous())
    z1 = e1[rows_b, t_idx] + p1[rows_r, u_idx]
    zg = eg[rows_b, t_idx] + pg[rows_r, u_idx]
    h = torch.tanh(z1) * torch.sigmoid(zg)
    logits = ops.linear(h.reshape(-1, H), model.fc2.weight, model.fc2.bias)
    # one non-zero per live row: seq_grad at the emitted symbol, blank entries scaled by 1/T (:225-233)
    val = seq_grad.reshape(-1, 1).expand(-1, S) * live
    val = torch.where(is_blk, val / float(T), val)
    surrogate = RiskFn.apply(logits, sym.reshape(-1).int(), val.reshape(-1).float().contiguous(), sm_scale)
    surrogate.backward()
    return float(surrogate.detach())
