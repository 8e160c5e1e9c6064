"""Generates tests/golden/bmuf_ws2.npz by running the REFERENCE BmufTrainer
(/root/reference/trainer/bmuf.py) itself: 2 processes, gloo on CPU.  Shims, all outside the
reference: `init_process_group(backend="nccl")` is redirected to gloo and `Tensor.cuda` is the
identity.  Only runs where /root/reference exists (this container); the vectors travel.
    python tests/golden/make_bmuf_golden.py
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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    real_init = dist.init_process_group
    dist.init_process_group = lambda backend=None, init_method=None, **kw: real_init(
        backend="gloo", init_method=init_method, **kw)
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, "/root/reference")
    from trainer.bmuf import BmufTrainer  # the reference class, unmodified
    model = C.make_model(rank)
    init = C.flat(model)
    tr = BmufTrainer(0, rank, world, model, C.BM, C.BLR)
    after_init = C.flat(model)
    rounds = []
    for rnd in range(C.ROUNDS):
        C.local_step(model, rank, rnd)
        assert tr.update_and_sync() == 1
        rounds.append(C.flat(model))
    t = torch.tensor([1.5 + rank, 10.0 * (rank + 1)])
    tr.sum_reduce(t)
    tr.broadcast(t)
    np.savez(out % rank, init=init, after_init=after_init, rounds=np.stack(rounds), loss=t.numpy())
    dist.destroy_process_group()


if __name__ == "__main__":
    world, port = 2, C.free_port()
    tmp = os.path.join(HERE, "_bmuf_rank%d.npz")
    mp.spawn(worker, args=(world, port, tmp), nprocs=world, join=True)
    z = [np.load(tmp % r) for r in range(world)]
    assert np.array_equal(z[0]["rounds"], z[1]["rounds"]), "reference ranks disagree"
    sim = C.simulate_reference_math(world)
    print("max |reference - single-process restatement| =", np.abs(sim - z[0]["rounds"]).max())
    np.savez_compressed(os.path.join(HERE, "bmuf_ws2.npz"), init0=z[0]["init"], init1=z[1]["init"],
                        after_init=z[0]["after_init"], rounds=z[0]["rounds"], loss=z[0]["loss"])
    for r in range(world):
        os.remove(tmp % r)
    print("wrote bmuf_ws2.npz")
