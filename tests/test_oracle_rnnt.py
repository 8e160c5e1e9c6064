"""Pins the CPU restatement of the RNN-T loss (oracle/rnnt_loss_ref.c).

The reference's loss lives in un-vendored `warp_rnnt`, and the reference holds no golden
vector for it, so the oracle is pinned (a) against the known-answer test that
warp_rnnt / warp-transducer publish in their own test suites (tests/golden/rnnt_kat.npz: cost AND
gradient, literals of the upstream tests) and (b) from first principles: exhaustive path enumeration,
finite differences, alpha/beta consistency, and the committed fixture tests/golden/rnnt_loss_small.npz
(made by tests/golden/make_rnnt_golden.py).
"""
import os

import numpy as np
import pytest

from oracle import rnnt as O
from helpers import make_case

GOLD = os.path.join(os.path.dirname(__file__), "golden", "rnnt_loss_small.npz")
KAT = os.path.join(os.path.dirname(__file__), "golden", "rnnt_kat.npz")


def test_upstream_known_answer_cost_and_gradient():
    """warp-transducer tests/test_cpu.cpp::small_test == warp_rnnt's test: acts (1,2,3,5), labels [1,2] -> cost
    4.495666 and d cost / d acts (the loss is applied to log_softmax(acts)); both oracle precisions."""
    z = np.load(KAT)
    acts = z["acts"].astype(np.float64)
    lp = acts - np.log(np.exp(acts).sum(-1, keepdims=True))
    for dtype, tol in ((np.float64, 2e-7), (np.float32, 1e-6)):
        costs, g = O.rnnt_loss(lp.astype(np.float32), z["labels"], z["frames_lengths"], z["labels_lengths"], dtype=dtype)
        assert abs(costs[0] - z["cost"][0]) < 1e-6                 # published to 7 significant digits
        # chain rule through log_softmax: d/d acts = g - softmax * sum_v g
        g = g.astype(np.float64)
        d_acts = g - np.exp(lp) * g.sum(-1, keepdims=True)
        assert np.abs(d_acts - z["grads_wrt_acts"]).max() < tol + 2e-7


@pytest.mark.parametrize("T,U,V,seed", [(1, 0, 3, 0), (1, 2, 4, 1), (3, 0, 4, 2), (4, 3, 5, 3),
                                        (5, 2, 7, 4), (2, 4, 3, 5)])
def test_cost_equals_brute_force_path_sum(T, U, V, seed):
    lp, y, tl, ul = make_case(1, T, U, V, seed)
    costs, _ = O.rnnt_loss(lp, y, tl, ul)
    bf = O.brute_force_cost(lp[0].astype(np.float64), y[0])
    assert abs(costs[0] - bf) < 1e-10


def test_gradient_matches_finite_differences_fp64():
    lp, y, tl, ul = make_case(2, 4, 3, 5, 7, ragged=False)
    tl[1], ul[1] = 3, 2
    y[1, 2:] = 5
    costs, g = O.rnnt_loss(lp, y, tl, ul)
    # fp64 perturbation needs fp64 inputs: go through brute force on the sub-lattice
    eps = 1e-6
    for n in range(2):
        Tn, Un = tl[n], ul[n]
        base = lp[n, :Tn, :Un + 1].astype(np.float64)
        for (t, u, v) in [(0, 0, 0), (0, 0, int(y[n, 0])), (Tn - 1, Un, 0), (1, 1, int(y[n, 1])),
                          (1, 1, 0), (1, 0, 3)]:
            p = base.copy(); p[t, u, v] += eps
            m = base.copy(); m[t, u, v] -= eps
            fd = (O.brute_force_cost(p, y[n, :Un]) - O.brute_force_cost(m, y[n, :Un])) / (2 * eps)
            assert abs(fd - g[n, t, u, v]) < 1e-7, (n, t, u, v, fd, g[n, t, u, v])
    # outside the (Tn, Un+1) sub-lattice the gradient is exactly zero
    assert np.all(g[1, 3:] == 0) and np.all(g[1, :, 3:] == 0)


def test_alpha_beta_agree_and_occupancy_sums():
    lp, y, tl, ul = make_case(3, 9, 5, 11, 11, ragged=True)
    costs, g, a, b = O.rnnt_loss(lp, y, tl, ul, want_lattice=True)
    for n in range(3):
        Tn, Un = tl[n], ul[n]
        ll_alpha = a[n, Tn - 1, Un] + lp[n, Tn - 1, Un, 0]
        assert abs(ll_alpha + costs[n]) < 1e-9
        # every path crosses each anti-diagonal once: occupancies on a diagonal sum to 1
        occ = np.exp(a[n, :Tn, :Un + 1] + b[n, :Tn, :Un + 1] + costs[n])
        for d in range(Tn + Un):
            s = sum(occ[d - u, u] for u in range(Un + 1) if 0 <= d - u < Tn)
            assert abs(s - 1) < 1e-9
        # total emitted mass: exactly Un labels and Tn blanks per path
        assert abs(-g[n].sum() - (Tn + Un)) < 1e-8
    assert np.all(costs > 0)


def test_padding_labels_are_never_read():
    lp, y, tl, ul = make_case(4, 6, 4, 9, 13, ragged=True)
    c1, g1 = O.rnnt_loss(lp, y, tl, ul)
    y2 = y.copy()
    for n in range(4):
        y2[n, ul[n]:] = 10 ** 6  # would segfault / change results if ever used as an index
    c2, g2 = O.rnnt_loss(lp, y2, tl, ul)
    assert np.array_equal(c1, c2) and np.array_equal(g1, g2)


def test_fp32_budget_against_fp64():
    lp, y, tl, ul = make_case(2, 40, 12, 33, 17, ragged=True)
    c64, g64 = O.rnnt_loss(lp, y, tl, ul, dtype=np.float64)
    c32, g32 = O.rnnt_loss(lp, y, tl, ul, dtype=np.float32)
    assert np.allclose(c32, c64, rtol=1e-5)
    assert np.abs(g32 - g64).max() < 1e-4


def test_bad_lengths_rejected():
    lp, y, tl, ul = make_case(1, 3, 2, 4, 0)
    with pytest.raises(ValueError):
        O.rnnt_loss(lp, y, [4], ul)
    with pytest.raises(ValueError):
        O.rnnt_loss(lp, y, tl, [3])


def test_golden_fixture():
    z = np.load(GOLD)
    costs, g = O.rnnt_loss(z["log_probs"], z["labels"], z["frames_lengths"], z["labels_lengths"])
    assert np.allclose(costs, z["costs"], rtol=0, atol=1e-12)
    assert np.allclose(g, z["grads"], rtol=0, atol=1e-12)
    # the fixture's own first-principles anchor
    for n in range(len(costs)):
        Tn, Un = z["frames_lengths"][n], z["labels_lengths"][n]
        bf = O.brute_force_cost(z["log_probs"][n, :Tn, :Un + 1].astype(np.float64), z["labels"][n, :Un])
        assert abs(bf - z["costs"][n]) < 1e-10
        assert abs(z["brute_force_costs"][n] - bf) < 1e-12
