"""Generates tests/golden/rnnt_loss_small.npz.

The reference holds no golden vector for the RNN-T loss (its arithmetic is in un-vendored
warp_rnnt), so the fixture is anchored on exhaustive path enumeration: `brute_force_costs`
is computed by pure-Python enumeration of every alignment, `costs`/`grads` by the fp64
oracle.  Run from the repo root:  python tests/golden/make_rnnt_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import rnnt as O  # noqa: E402
from helpers import make_case  # noqa: E402

lp, y, tl, ul = make_case(4, 6, 4, 8, seed=20260925, ragged=True)
costs, grads, alphas, betas = O.rnnt_loss(lp, y, tl, ul, want_lattice=True)
bf = np.array([O.brute_force_cost(lp[n, :tl[n], :ul[n] + 1].astype(np.float64), y[n, :ul[n]])
               for n in range(4)])
assert np.abs(bf - costs).max() < 1e-10
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rnnt_loss_small.npz")
np.savez_compressed(out, log_probs=lp, labels=y, frames_lengths=tl, labels_lengths=ul, costs=costs,
                    grads=grads, alphas=alphas, betas=betas, brute_force_costs=bf)
print("wrote", out, os.path.getsize(out), "bytes")
