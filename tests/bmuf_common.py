"""Shared scenario for the BMUF tests and for tests/golden/make_bmuf_golden.py:
2 workers, a small MLP with rank-dependent initial weights, 3 blocks of fake local SGD."""
import socket

import numpy as np
import torch
import torch.nn as nn

ROUNDS = 3
BM, BLR = 0.9, 1.0


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_model(rank):
    torch.manual_seed(1000 + rank)  # different per rank: the initial broadcast must fix that
    return nn.Sequential(nn.Linear(7, 5), nn.ReLU(), nn.Linear(5, 3), nn.BatchNorm1d(3))


def local_step(model, rank, rnd):
    """Stand-in for sync_period steps of local SGD: a seeded perturbation per rank and round."""
    g = torch.Generator().manual_seed(77 * (rank + 1) + rnd)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.05)


def flat(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy().copy()


def simulate_reference_math(world):
    """Single-process restatement of bmuf.py:76-98 on the same scenario (fp32, PyTorch ops)."""
    models = [make_model(r) for r in range(world)]
    G = torch.from_numpy(flat(models[0]))
    for m in models:  # broadcast + _copy_vec_to_param
        torch.nn.utils.vector_to_parameters(G.clone(), m.parameters())
    dprev = torch.zeros_like(G)
    out = []
    for rnd in range(ROUNDS):
        for r, m in enumerate(models):
            local_step(m, r, rnd)
        delta = sum((G - torch.from_numpy(flat(m))) for m in models)
        delta = delta / float(world)
        dprev = BM * dprev + (BLR * (1 - BM) * delta)
        G = G - (1 + BM) * dprev
        for m in models:
            torch.nn.utils.vector_to_parameters(G.clone(), m.parameters())
        out.append(G.numpy().copy())
    return np.stack(out)


# ---- BMUF-Adam scenario (trainer/bmuf.py:191-333) -----------------------------------------------------
SYNC_PERIOD = 3


def make_adam(model):
    return torch.optim.Adam(model.parameters(), lr=1e-2, betas=(0.9, 0.98))


def local_adam_steps(model, optim, rank, rnd):
    """sync_period local Adam steps on seeded fake gradients (every parameter gets one, so the optimizer
    state the trainer exchanges exists)."""
    for k in range(SYNC_PERIOD):
        g = torch.Generator().manual_seed(1000 * (rank + 1) + 10 * rnd + k)
        for p in model.parameters():
            p.grad = torch.randn(p.shape, generator=g) * 0.1
        optim.step()


def adam_state(model, optim):
    """Flat (params, exp_avg, exp_avg_sq, steps) snapshot."""
    ps = [p for p in model.parameters()]
    st = [optim.state[p] for p in ps]
    return (flat(model),
            torch.cat([s["exp_avg"].reshape(-1) for s in st]).numpy().copy(),
            torch.cat([s["exp_avg_sq"].reshape(-1) for s in st]).numpy().copy(),
            np.array([float(s["step"]) for s in st]))
