"""BMUF (block-momentum model averaging) over RCCL/xGMI -- host side.

Mirrors `trainer.bmuf.BmufTrainer` (/root/reference/trainer/bmuf.py:38-109): same constructor,
`update_and_sync() -> SUCCESS|STOP`, `sum_reduce(t)`, `broadcast(t)`; same block update
(bmuf.py:93-96).  Re-designed for 8 x MI355X on one node:

* parameters are re-pointed to views of ONE flat fp32 vector, so `parameters_to_vector`
  (bmuf.py:63,84) and `_copy_vec_to_param` (bmuf.py:14-35, one copy kernel per tensor) vanish;
* `reduce(dst=master)` + master-only update + `broadcast` (two serialised 361 MB collectives
  through rank 0's links, bmuf.py:87,97) become ONE all-reduce(SUM) of `delta`, after which
  every rank applies the identical update (all ranks hold `delta_prev`; inputs are bitwise
  identical after the all-reduce, so the replicas stay bitwise identical);
* the elementwise math is two single-pass HIP kernels (include/pika_bmuf.h);
* the NaN guard (bmuf.py:89) reads a device flag -- with an all-reduce every rank sees the
  same sum, so ranks can no longer disagree about STOP.

The collective layer is device-agnostic `torch.distributed` (backend "nccl" == RCCL on ROCm;
"gloo" when the model lives on the CPU, which is how the world_size-2 tests run without a
GPU).  On HIP tensors the fused kernels are mandatory: a missing library raises.
"""
import os

import torch
import torch.distributed as dist

from . import _lib

SUCCESS = 1
STOP = 0


def _stream():
    return torch.cuda.current_stream().cuda_stream


class BmufTrainer(object):
    """Nesterov block-momentum BMUF trainer (drop-in for trainer.bmuf.BmufTrainer)."""

    def __init__(self, master_node, rank, world_size, model, block_momentum, block_lr,
                 backend=None):
        self.master_node = master_node
        self.rank = rank
        self.world_size = world_size
        self.model = model
        self.block_momentum = float(block_momentum)
        self.block_lr = float(block_lr)
        params = [p for p in model.parameters()]
        if not params:
            raise ValueError("BmufTrainer: model has no parameters")
        dev = params[0].device
        self.is_hip = dev.type == "cuda"
        if not dist.is_initialized():
            # same call as bmuf.py:59; "nccl" is RCCL on ROCm.  CPU models use gloo.
            dist.init_process_group(backend=backend or ("nccl" if self.is_hip else "gloo"),
                                    init_method="env://")
        if self.is_hip:
            _lib.lib()  # fail loudly now, not at the first sync
        n = sum(p.numel() for p in params)
        # flat local vector; parameters become views of it (order = model.parameters() order,
        # which defines the BMUF vector in the reference too)
        self.local = torch.empty(n, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            if p.dtype != torch.float32:
                raise TypeError("BmufTrainer expects fp32 master parameters, got %s" % p.dtype)
            k = p.numel()
            self.local[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.local[off:off + k].view(p.shape)
            off += k
        self.param = self.local.clone()                 # global model G        (bmuf.py:64)
        dist.broadcast(self.param, src=master_node)     # initial sync           (bmuf.py:66)
        self.local.copy_(self.param)                    # every rank starts at G (bmuf.py:74)
        self.delta_prev = torch.zeros_like(self.param)  # on EVERY rank (reference: master only)
        self.delta = torch.empty_like(self.param)
        self._flag = torch.zeros(1, dtype=torch.int32, device=dev)
        if self.is_hip:
            self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._flag_event = torch.cuda.Event()
        self._stop_pending = False
        # True (default): update_and_sync() returns STOP for the block whose summed delta holds a NaN -- the reference's
        # timing (bmuf.py:89-90), at the price of one blocking 4-byte read per block (the reference loop reads the loss
        # every step anyway).  PIKA_BMUF_SYNC_STOP=0: the host learns of it at the NEXT call (the device has already
        # skipped the update); a loop that uses it must ask `pending_stop()` after its last block and before a checkpoint
        self.sync_stop = os.environ.get("PIKA_BMUF_SYNC_STOP", "1") == "1"
        self.collective_events = None     # set to [] to have every all-reduce bracketed by HIP events

    def _rebind_detached_parameters(self):
        """The reference re-reads the parameters at every sync (`parameters_to_vector`, bmuf.py:84).  Here they are
        views of `self.local`; anything that re-bound `p.data` since (model.to / .float(), load_state_dict(assign=True),
        flatten_parameters, module re-wrapping) would silently detach the model from the block update.  Adopt the
        parameter's CURRENT values and re-point it.  (~300 pointer compares per sync.)"""
        off, base, n_fixed = 0, self.local.data_ptr(), 0
        for p in self.model.parameters():
            k = p.numel()
            if p.data_ptr() != base + 4 * off or p.dtype != torch.float32 or p.device != self.local.device:
                self.local[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.local[off:off + k].view(p.shape)
                n_fixed += 1
            off += k
        if off != self.local.numel():
            raise RuntimeError("BmufTrainer: the model's parameter count changed since construction "
                               "(%d -> %d elements)" % (self.local.numel(), off))
        return n_fixed

    def pending_stop(self):
        """Deferred mode (PIKA_BMUF_SYNC_STOP=0) only: did the LAST block hold a NaN?  Blocks until its flag has arrived;
        clears the pending state (the update kernel left parameters, delta_prev and the local model untouched)."""
        if self.is_hip and self._stop_pending:
            self._flag_event.synchronize()
            self._stop_pending = False
            return bool(int(self._flag_host[0]))
        return False

    # -- the block update ----------------------------------------------------------------
    def update_and_sync(self):
        n = self.param.numel()
        if self.is_hip and self._stop_pending:      # the previous block's flag (its copy finished long ago)
            self._flag_event.synchronize()
            self._stop_pending = False
            if int(self._flag_host[0]):
                return STOP
        self._rebind_detached_parameters()
        if self.is_hip:
            lib = _lib.lib()
            with torch.cuda.device(self.param.device):
                _lib.check(lib.pika_bmuf_delta(self.param.data_ptr(), self.local.data_ptr(),
                                               self.delta.data_ptr(), n, _stream()), "pika_bmuf_delta")
        else:
            torch.sub(self.param, self.local, out=self.delta)
        ev = None
        if self.collective_events is not None and self.is_hip:      # bench.py: HIP events around the exchange step
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        dist.all_reduce(self.delta, op=dist.ReduceOp.SUM)
        if ev is not None:
            ev[1].record()
            self.collective_events.append(ev)
        world = float(self.world_size)
        bm, blr = self.block_momentum, self.block_lr
        if self.is_hip:
            # the NaN guard of bmuf.py:89-90 is decided ON THE DEVICE: the flag kernel runs, and the update kernel leaves
            # every vector untouched when the flag is set -- no blocking read between the all-reduce and the update.  The
            # host learns of it from a pinned copy of the flag: at once with sync_stop (the reference's timing), else at
            # the next call (every rank holds the same summed delta, so every rank stops at the same call)
            with torch.cuda.device(self.param.device):
                self._flag.zero_()
                _lib.check(lib.pika_bmuf_nan_flag(self.delta.data_ptr(), n, self._flag.data_ptr(), _stream()),
                           "pika_bmuf_nan_flag")
                _lib.check(lib.pika_bmuf_update(self.delta.data_ptr(), self.delta_prev.data_ptr(),
                                                self.param.data_ptr(), self.local.data_ptr(), n,
                                                world, bm, blr, self._flag.data_ptr(), _stream()), "pika_bmuf_update")
                self._flag_host.copy_(self._flag, non_blocking=True)
                self._flag_event.record()
            if self.sync_stop:
                self._flag_event.synchronize()
                if int(self._flag_host[0]):
                    return STOP
            else:
                self._stop_pending = True
            return SUCCESS
        if bool(torch.isnan(self.delta).any().item()):
            return STOP
        # device-agnostic restatement of the same four lines, used when the model is on the
        # CPU (gloo plumbing tests); float32 scalars rounded exactly as the kernel rounds them
        f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
        c = f32(f32(blr) * f32(1.0 - f32(bm)))
        self.delta.div_(world)                      # (bmuf.py:93: a division)
        self.delta_prev.mul_(f32(bm)).add_(self.delta * c)
        self.param.sub_(self.delta_prev * f32(1.0 + f32(bm)))
        self.local.copy_(self.param)
        return SUCCESS

    # -- small-tensor helpers used for the epoch loss (train_transducer_bmuf_otfaug.py:140-143)
    def broadcast(self, tensor):
        dist.broadcast(tensor=tensor, src=self.master_node, async_op=False)

    def sum_reduce(self, tensor):
        # reference: reduce(dst=master) then the caller broadcasts; an all-reduce leaves the
        # same value on the master and makes the following broadcast a no-op in effect
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)


def _flatten_parameters(params, dev):
    """Re-points the parameters to views of ONE flat fp32 vector (order = model.parameters()) and returns it."""
    n = sum(p.numel() for p in params)
    flat = torch.empty(n, dtype=torch.float32, device=dev)
    off = 0
    for p in params:
        if p.dtype != torch.float32:
            raise TypeError("BMUF expects fp32 master parameters, got %s" % p.dtype)
        k = p.numel()
        flat[off:off + k].copy_(p.data.reshape(-1))
        p.data = flat[off:off + k].view(p.shape)
        off += k
    return flat


class BmufAdamTrainer(object):
    """BMUF-Adam (Chen et al. 2020), drop-in for trainer.bmuf.BmufAdamTrainer (/root/reference/trainer/bmuf.py:191-333):
    same constructor (`..., block_momentum, block_lr, sync_period, optim`), `update_and_sync()`, `broadcast`,
    `sum_reduce`; the block update of the parameters AND of Adam's first / second moments, whose averaged,
    bias-compensated values are what the optimizer continues from (`state['step']` advanced by rho * block_momentum).

    Laid out for one node of MI355Xs like BmufTrainer above, not like the reference (which concatenates three freshly
    gathered vectors per block, reduces them to the master, updates there and broadcasts three vectors back):

    * ONE exchange buffer `xch` = [delta | exp_avg | exp_avg_sq] lives for the whole run.  The parameters are views of a
      flat local vector; the optimizer's moment tensors are re-pointed (at the first block, when they exist) to views of
      the second and third segment of `xch` -- Adam updates them in place, so at a block boundary the buffer already
      holds what has to travel: nothing is gathered, nothing is scattered back;
    * ONE in-place all-reduce(SUM) of `xch`, after which every rank applies the identical update (every rank keeps
      delta_prev and the block moments; replicas stay bitwise identical);
    * on a HIP device the update is two single-pass kernels (include/pika_bmuf.h: pika_bmuf_update for the parameters,
      pika_bmuf_adam_moments for both moments -- the summed state goes in, the block moments come out in the same
      memory) behind the device-side NaN flag of BmufTrainer; on the CPU (gloo tests) the same arithmetic as in-place
      torch ops on the segments.
    A NaN in the summed buffer returns STOP on every rank at the same block (the optimizer state is then the sum, not a
    usable state: STOP ends the run, as in the reference)."""

    def __init__(self, master_node, rank, world_size, model, block_momentum, block_lr, sync_period, optim,
                 backend=None):
        self.master_node, self.rank, self.world_size = master_node, rank, world_size
        self.model, self.optim = model, optim
        self.block_momentum, self.block_lr, self.sync_period = float(block_momentum), float(block_lr), sync_period
        params = [p for p in model.parameters()]
        dev = params[0].device
        self.is_hip = dev.type == "cuda"
        if not dist.is_initialized():
            dist.init_process_group(backend=backend or ("nccl" if self.is_hip else "gloo"), init_method="env://")
        if self.is_hip:
            _lib.lib()
        self.rho = 0.0
        self.betas = (0.9, 0.999)
        self._opt_params = []
        for group in optim.param_groups:
            self.betas = group['betas']
            self._opt_params += list(group['params'])
        self.local = _flatten_parameters(params, dev)
        self.param = self.local.clone()                      # the global model
        dist.broadcast(tensor=self.param, src=master_node, async_op=False)
        self.local.copy_(self.param)
        self.num_param = n = self.param.numel()
        self.dim = dim = sum(p.numel() for p in self._opt_params)
        self.xch = torch.zeros(n + 2 * dim, dtype=torch.float32, device=dev)       # [delta | exp_avg | exp_avg_sq]
        self.delta_prev = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(dim, dtype=torch.float32, device=dev)           # the block moments
        self.exp_avg_sq = torch.zeros(dim, dtype=torch.float32, device=dev)
        self._flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self._adopted = False

    def _adopt_optimizer_state(self):
        """The optimizer's moment tensors become views of the exchange buffer (once: Adam creates them at its first step
        and updates them in place from then on).  Re-done when something re-bound them (load_state_dict)."""
        n, dim, off = self.num_param, self.dim, 0
        base1, base2 = self.xch[n:n + dim], self.xch[n + dim:]
        for p in self._opt_params:
            k = p.numel()
            st = self.optim.state.get(p)
            if st and 'exp_avg' in st:
                for name, seg in (('exp_avg', base1), ('exp_avg_sq', base2)):
                    t = st[name]
                    view = seg[off:off + k]
                    if t.data_ptr() != view.data_ptr():
                        view.copy_(t.reshape(-1))
                        st[name] = view.view(t.shape)
            else:       # no optimizer step has touched this parameter yet: zero moments travel for it
                base1[off:off + k].zero_()
                base2[off:off + k].zero_()
            off += k
        self._adopted = True

    def update_and_sync(self):
        n, dim, W, bm = self.num_param, self.dim, float(self.world_size), self.block_momentum
        self._adopt_optimizer_state()
        delta, m_sum, v_sum = self.xch[:n], self.xch[n:n + dim], self.xch[n + dim:]
        lib = _lib.lib() if self.is_hip else None
        if self.is_hip:
            with torch.cuda.device(self.xch.device):
                _lib.check(lib.pika_bmuf_delta(self.param.data_ptr(), self.local.data_ptr(), delta.data_ptr(), n, _stream()),
                           "pika_bmuf_delta")
        else:
            torch.sub(self.param, self.local, out=delta)
        dist.all_reduce(self.xch, op=dist.ReduceOp.SUM)
        # the coefficients of this block (float64 on the host, like the reference's Python floats)
        rho = bm * self.rho + self.sync_period
        b1t, b2t = self.betas[0] ** self.sync_period, self.betas[1] ** self.sync_period
        b1r, b2r = self.betas[0] ** (rho * bm), self.betas[1] ** (rho * bm)
        c_avg = (b1t * (b1r - 1), 1 - b1t * b1r, 1 - b1t)
        c_sq = (b2t * (b2r - 1), 1 - b2t * b2r, 1 - b2t)
        if self.is_hip:
            with torch.cuda.device(self.xch.device):
                self._flag.zero_()
                _lib.check(lib.pika_bmuf_nan_flag(self.xch.data_ptr(), self.xch.numel(), self._flag.data_ptr(), _stream()),
                           "pika_bmuf_nan_flag")
                _lib.check(lib.pika_bmuf_update(delta.data_ptr(), self.delta_prev.data_ptr(), self.param.data_ptr(),
                                                self.local.data_ptr(), n, W, bm, self.block_lr,
                                                self._flag.data_ptr(), _stream()), "pika_bmuf_update")
                _lib.check(lib.pika_bmuf_adam_moments(m_sum.data_ptr(), self.exp_avg.data_ptr(), v_sum.data_ptr(),
                                                      self.exp_avg_sq.data_ptr(), dim, W, *c_avg, *c_sq,
                                                      self._flag.data_ptr(), _stream()), "pika_bmuf_adam_moments")
            if int(self._flag.item()):
                return STOP
        else:
            if bool(torch.isnan(self.xch).any().item()):
                return STOP
            self.xch.div_(W)
            self.delta_prev.mul_(bm).add_(delta * (self.block_lr * (1 - bm)))
            self.param.sub_(self.delta_prev * (1 + bm))
            self.local.copy_(self.param)
            for blk, x, c in ((self.exp_avg, m_sum, c_avg), (self.exp_avg_sq, v_sum, c_sq)):
                blk.mul_(c[0]).add_(x * c[1]).div_(c[2])
                x.copy_(blk)                  # the optimizer's own tensors: it continues from the block moments
        self.rho = rho
        for p in self._opt_params:
            st = self.optim.state.get(p)
            if st and 'step' in st:
                st['step'] += self.rho * bm
        return SUCCESS

    def broadcast(self, tensor):
        dist.broadcast(tensor=tensor, src=self.master_node, async_op=False)

    def sum_reduce(self, tensor):
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
