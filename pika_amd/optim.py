"""Per-step optimizer passes of the training scripts on the device (include/pika_optim.h):

    torch.nn.utils.clip_grad_norm_(model.parameters(), args.grad_clip, norm_type=inf)
    optimizer.step()                      # optim.SGD(model.parameters(), lr, momentum=..., nesterov=True)

(trainer/train_transducer_bmuf_otfaug.py:105-110, :53-55, :121-123).  `install()` -- called by pika_amd.launch --
makes exactly these two calls of an UNCHANGED script take three HIP launches over all ~300 parameter tensors instead
of torch's ~10 multi-tensor launches; anything else (other norm types, other optimizers or SGD options, CPU tensors,
closures) goes to the stock implementation.  Results: same values as torch (every product rounded separately, no
FMA contraction); momentum buffers live in `state[p]["momentum_buffer"]` as torch keeps them.
"""
import ctypes

import torch

from . import _lib

CHUNK = 16384
_vp, _i, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _Tables(object):
    """Device chunk tables for a fixed list of tensor sizes + pinned staging for the pointer tables (gradients are
    re-allocated every step by zero_grad(set_to_none=True), so their addresses are uploaded per call)."""

    def __init__(self, numels, device):
        ct, co, cl = [], [], []
        for t, n in enumerate(numels):
            for off in range(0, n, CHUNK):
                ct.append(t)
                co.append(off)
                cl.append(min(CHUNK, n - off))
        self.n_chunks, self.n = len(ct), len(numels)
        self.ct = torch.tensor(ct, dtype=torch.int32, device=device)
        self.co = torch.tensor(co, dtype=torch.int64, device=device)
        self.cl = torch.tensor(cl, dtype=torch.int32, device=device)
        # pinned staging: a ring of RING generations of the three pointer rows.  The copy out of a row is asynchronous,
        # and a host that runs ahead of the device (no .item() between optimizer steps) must not rewrite a row whose copy
        # has not executed yet: a row is re-used only after the event recorded behind its last copy has completed.
        self.host = torch.empty(self.RING, 3, self.n, dtype=torch.int64).pin_memory()
        self.host_np = self.host.numpy()            # element writes through torch cost ~5 us each; numpy: one call
        self.events = [[None] * 3 for _ in range(self.RING)]
        self.gen = [0, 0, 0]
        self.dev = torch.empty(3, self.n, dtype=torch.int64, device=device)
        self.norm = torch.zeros(1, dtype=torch.float32, device=device)
        self.last = [None, None, None]              # what each device row holds: an unchanged table is not uploaded again

    RING = 8

    def upload(self, row, tensors):
        if torch.cuda.is_current_stream_capturing():
            # lr, momentum and the `first` flag are host scalars of the launch and the pointer rows are host memory a
            # captured copy would re-read at every replay: a captured optimizer step would silently freeze all of them.
            # The graphed training step (pika_amd/train_graph.py) captures the model's forward / backward only.
            raise RuntimeError("pika_amd.optim: clip / SGD steps are eager launches (3 per step); do not capture them")
        ptrs = [t.data_ptr() for t in tensors]
        if self.last[row] == ptrs:      # parameters and momentum buffers stay put; so do the gradients of a replayed step
            return self.dev[row].data_ptr()
        self.last[row] = ptrs
        g = self.gen[row] = (self.gen[row] + 1) % self.RING
        ev = self.events[g][row]
        if ev is not None:
            ev.synchronize()
        self.host_np[g, row, :] = ptrs
        self.dev[row].copy_(self.host[g, row], non_blocking=True)
        if ev is None:
            ev = self.events[g][row] = torch.cuda.Event()
        ev.record()
        return self.dev[row].data_ptr()


_CACHE = {}


def _tables(tensors):
    key = (tensors[0].device.index,) + tuple(t.numel() for t in tensors)
    tb = _CACHE.get(key)
    if tb is None:
        tb = _CACHE[key] = _Tables([t.numel() for t in tensors], tensors[0].device)
    return tb


def _fusable(tensors):
    return bool(tensors) and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors) \
        and len({t.device for t in tensors}) == 1


def clip_grad_inf_norm_(parameters, max_norm):
    """clip_grad_norm_(parameters, max_norm, norm_type=inf) for fp32 HIP gradients; returns the total norm (0-dim)."""
    grads = [p.grad for p in parameters if p.grad is not None]
    tb = _tables(grads)
    lib = _lib.lib()
    with torch.cuda.device(grads[0].device):
        gp = tb.upload(1, grads)
        _lib.check(lib.pika_multi_absmax(gp, tb.ct.data_ptr(), tb.co.data_ptr(), tb.cl.data_ptr(), tb.n_chunks,
                                         tb.norm.data_ptr(), _stream()), "pika_multi_absmax")
        _lib.check(lib.pika_multi_scale_by_clip(gp, tb.ct.data_ptr(), tb.co.data_ptr(), tb.cl.data_ptr(), tb.n_chunks,
                                                tb.norm.data_ptr(), float(max_norm), _stream()),
                   "pika_multi_scale_by_clip")
    return tb.norm[0].clone()


_torch_clip = torch.nn.utils.clip_grad_norm_
_TorchSGD = torch.optim.SGD


def clip_grad_norm_(parameters, max_norm, norm_type=2.0, error_if_nonfinite=False, foreach=None):
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    parameters = list(parameters)
    grads = [p.grad for p in parameters if p.grad is not None]
    if float(norm_type) == float("inf") and not error_if_nonfinite and _fusable(grads):
        return clip_grad_inf_norm_(parameters, max_norm)
    return _torch_clip(parameters, max_norm, norm_type, error_if_nonfinite=error_if_nonfinite, foreach=foreach)


class SGD(_TorchSGD):
    """torch.optim.SGD whose Nesterov-momentum step over fp32 HIP parameters is ONE launch."""

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or len(self.param_groups) != 1:
            return super().step(closure)
        g = self.param_groups[0]
        params = [p for p in g["params"] if p.grad is not None]
        ok = (g["nesterov"] and g["momentum"] > 0 and g["dampening"] == 0 and g["weight_decay"] == 0
              and not g.get("maximize", False) and len(params) == len(g["params"])
              and _fusable([p.data for p in params]) and _fusable([p.grad for p in params])
              and not any(p.grad.is_sparse for p in params))
        if not ok:
            return super().step(closure)
        first = "momentum_buffer" not in self.state[params[0]] or self.state[params[0]]["momentum_buffer"] is None
        if first:
            flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=params[0].device)
            off = 0
            for p in params:
                self.state[p]["momentum_buffer"] = flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
        bufs = [self.state[p]["momentum_buffer"] for p in params]
        tb = _tables(params)
        lib = _lib.lib()
        with torch.cuda.device(params[0].device):
            pp, gp, bp = tb.upload(0, params), tb.upload(1, [p.grad for p in params]), tb.upload(2, bufs)
            _lib.check(lib.pika_multi_sgd_nesterov(pp, gp, bp, tb.ct.data_ptr(), tb.co.data_ptr(), tb.cl.data_ptr(),
                                                   tb.n_chunks, float(g["lr"]), float(g["momentum"]), int(first),
                                                   _stream()), "pika_multi_sgd_nesterov")
        return None


def install():
    """Route the two calls through the kernels above (idempotent).  `SGD` subclasses the stock class, so
    isinstance checks and state_dict round trips keep working."""
    torch.nn.utils.clip_grad_norm_ = clip_grad_norm_
    torch.optim.SGD = SGD


def uninstall():
    torch.nn.utils.clip_grad_norm_ = _torch_clip
    torch.optim.SGD = _TorchSGD
