"""Generates tests/golden/bmuf_adam_ws2.npz by running the REFERENCE BmufAdamTrainer
(/root/reference/trainer/bmuf.py:191-333) itself: 2 processes, gloo on CPU, torch.optim.Adam.  Same shims as
make_bmuf_golden.py (nccl -> gloo, Tensor.cuda = identity), all outside the reference.
    python tests/golden/make_bmuf_adam_golden.py
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bmuf_common as C  # noqa: E402


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    real_init = dist.init_process_group
    dist.init_process_group = lambda backend=None, init_method=None, **kw: real_init(
        backend="gloo", init_method=init_method, **kw)
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, "/root/reference")
    from trainer.bmuf import BmufAdamTrainer  # the reference class, unmodified
    model = C.make_model(rank)
    optim = C.make_adam(model)
    tr = BmufAdamTrainer(0, rank, world, model, C.BM, C.BLR, C.SYNC_PERIOD, optim)
    snaps = []
    for rnd in range(C.ROUNDS):
        C.local_adam_steps(model, optim, rank, rnd)
        assert tr.update_and_sync() == 1
        snaps.append(C.adam_state(model, optim))
    np.savez(out % rank, params=np.stack([s[0] for s in snaps]), exp_avg=np.stack([s[1] for s in snaps]),
             exp_avg_sq=np.stack([s[2] for s in snaps]), steps=np.stack([s[3] for s in snaps]), rho=tr.rho)
    dist.destroy_process_group()


if __name__ == "__main__":
    world, port = 2, C.free_port()
    tmp = os.path.join(HERE, "_bmufadam_rank%d.npz")
    mp.spawn(worker, args=(world, port, tmp), nprocs=world, join=True)
    z = [np.load(tmp % r) for r in range(world)]
    for k in ("params", "exp_avg", "exp_avg_sq", "steps"):
        assert np.array_equal(z[0][k], z[1][k]), "reference ranks disagree on " + k
    np.savez_compressed(os.path.join(HERE, "bmuf_adam_ws2.npz"), **{k: z[0][k] for k in z[0].files})
    for r in range(world):
        os.remove(tmp % r)
    print("wrote bmuf_adam_ws2.npz; final steps", z[0]["steps"][-1][:3], "rho", z[0]["rho"])
