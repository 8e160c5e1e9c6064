"""One training step -- forward, RNN-T loss, backward, inf-norm clip, Nesterov SGD -- as ONE hipGraph.

The eager step of the reference loop (trainer/train_transducer_bmuf_otfaug.py:95-110) is ~650 kernel launches issued
from Python: 42 ms of host time for 44 ms of device time on an idle 128-core box (profiles/r2_train_step_host_bound.txt),
which is exactly where eight ranks on one host lose their scaling.  Captured once, the same launch sequence costs the
host one graph launch per step and the device no launch gaps.

What makes the sequence replayable:
  * fixed shapes: the batch lives in static device buffers the caller's tensors are copied into (the bucketed loader of
    the recipes yields a handful of shapes; one graph per (B, T, U) shape is kept);
  * dropout: every kernel of libpika_amd.so that takes a dropout seed adds a device word to it
    (pika_set_dropout_salt); the word is re-drawn on the device before every replay, so each replay has new masks
    and the forward / backward of one replay agree; torch's own dropout (the few eager-fallback call sites) is
    graph-safe by itself (philox offsets are graph inputs);
  * the optimizer: momentum buffers and gradients live at fixed addresses inside the graph's memory pool; the reference
    re-creates the optimizer after every BMUF block (:121) = momentum buffers restart from zero, which is
    `reset_momentum()` here (buf = 0.9 * 0 + g equals the first step of a fresh torch.optim.SGD);
  * nothing on the path reads the device (the loss value stays in a static tensor; the caller decides when to read it).
The first `warmup` calls run eagerly (they are real training steps: lazy workspaces, kernel attributes and the
allocator reach their steady state), the next call captures and replays.
"""
import torch

from . import _lib


class GraphedTrainStep(object):
    def __init__(self, model, loss_fn, make_optimizer, clip=3.0, warmup=2):
        self.model, self.loss_fn, self.clip = model, loss_fn, float(clip)
        self.optimizer = make_optimizer()
        self.warmup, self.calls = int(warmup), 0
        self.graphs = {}            # (data shape, labels shape) -> (graph, static buffers, static loss)
        dev = next(model.parameters()).device
        self.salt = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().pika_set_dropout_salt(self.salt.data_ptr()), "pika_set_dropout_salt")

    def close(self):
        _lib.lib().pika_set_dropout_salt(None)

    def _body(self, data, labels, len_b, ali):
        out = self.model(data, labels.long(), len_b, True)
        loss = self.loss_fn(out, labels.int(), len_b, ali).sum()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip, norm_type=float("inf"))
        self.optimizer.step()
        return loss

    def reset_momentum(self):
        """What re-creating the optimizer does to its state (train_transducer_bmuf_otfaug.py:121), at fixed addresses."""
        bufs = [st["momentum_buffer"] for st in self.optimizer.state.values() if st.get("momentum_buffer") is not None]
        if bufs:
            torch._foreach_zero_(bufs)

    def _capture(self, data, labels, len_b, ali):
        static = [t.clone() for t in (data, labels, len_b, ali)]
        g = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)      # gradients are (re)allocated inside the graph's pool
        # thread_local: the loader thread keeps issuing its own uploads / kernels on its side stream during the capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            loss = self._body(*static)
        entry = (g, static, loss.detach())
        self.graphs[(tuple(data.shape), tuple(labels.shape))] = entry
        return entry

    def __call__(self, data, labels, len_b, ali):
        """data (B,T,F) f32 after CMVN / SpecAugment, labels (B,U), len_b / ali (B,) int32: one optimisation step;
        returns the summed loss as a device tensor (static across calls: read or clone it before the next call)."""
        self.calls += 1
        self.salt.random_()                              # device-side draw: no host involvement
        if self.calls <= self.warmup:
            self.optimizer.zero_grad(set_to_none=True)
            return self._body(data, labels, len_b, ali).detach()
        entry = self.graphs.get((tuple(data.shape), tuple(labels.shape)))
        fresh = entry is None
        if fresh:
            entry = self._capture(data, labels, len_b, ali)     # records, does not run: replayed below on this batch
        g, static, loss = entry
        if not fresh:
            for s, t in zip(static, (data, labels, len_b, ali)):
                s.copy_(t, non_blocking=True)
        g.replay()
        return loss
