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

    # -- the block update ----------------------------------------------------------------
    def update_and_sync(self):
        n = self.param.numel()
        if self.is_hip:
            lib = _lib.lib()
            with torch.cuda.device(self.param.device):
                _lib.check(lib.pika_bmuf_delta(self.param.data_ptr(), self.local.data_ptr(),
                                               self.delta.data_ptr(), n, _stream()), "pika_bmuf_delta")
        else:
            torch.sub(self.param, self.local, out=self.delta)
        dist.all_reduce(self.delta, op=dist.ReduceOp.SUM)
        if self._has_nan():
            return STOP
        inv_world = 1.0 / float(self.world_size)
        bm, blr = self.block_momentum, self.block_lr
        if self.is_hip:
            with torch.cuda.device(self.param.device):
                _lib.check(lib.pika_bmuf_update(self.delta.data_ptr(), self.delta_prev.data_ptr(),
                                                self.param.data_ptr(), self.local.data_ptr(), n,
                                                inv_world, bm, blr, _stream()), "pika_bmuf_update")
        else:
            # device-agnostic restatement of the same four lines, used when the model is on the
            # CPU (gloo plumbing tests); float32 scalars rounded exactly as the kernel rounds them
            f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
            c = f32(f32(blr) * f32(1.0 - f32(bm)))
            self.delta.mul_(f32(inv_world))
            self.delta_prev.mul_(f32(bm)).add_(self.delta * c)
            self.param.sub_(self.delta_prev * f32(1.0 + f32(bm)))
            self.local.copy_(self.param)
        return SUCCESS

    def _has_nan(self):
        if self.is_hip:
            lib = _lib.lib()
            self._flag.zero_()
            with torch.cuda.device(self.param.device):
                _lib.check(lib.pika_bmuf_nan_flag(self.delta.data_ptr(), self.delta.numel(),
                                                  self._flag.data_ptr(), _stream()), "pika_bmuf_nan_flag")
            return bool(self._flag.item())
        return bool(torch.isnan(self.delta).any().item())

    # -- small-tensor helpers used for the epoch loss (train_transducer_bmuf_otfaug.py:140-143)
    def broadcast(self, tensor):
        dist.broadcast(tensor=tensor, src=self.master_node, async_op=False)

    def sum_reduce(self, tensor):
        # reference: reduce(dst=master) then the caller broadcasts; an all-reduce leaves the
        # same value on the master and makes the following broadcast a no-op in effect
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
