"""ctypes/numpy front-end of oracle/rnnt_loss_ref.c (checker, not product).

PARITY UNPINNED at the warp_rnnt boundary (see the header of rnnt_loss_ref.c);
pinned by brute force + finite differences in tests/test_oracle_rnnt.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "rnnt_loss_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        for name in ("oracle_rnnt_loss_f64", "oracle_rnnt_loss_f32"):
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 4
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads():
    return int(lib().oracle_num_threads())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def rnnt_loss(log_probs, labels, frames_lengths, labels_lengths, blank=0,
              dtype=np.float64, want_grads=True, want_lattice=False):
    """costs (B,), grads (B,T,U1,V) | None, [alphas, betas (B,T,U1)].

    ``log_probs`` float32 (B,T,U1,V); ``labels`` int32 (B,U1-1).
    ``dtype`` selects the arithmetic: float64 = truth, float32 = tolerance budget.
    """
    lp = np.ascontiguousarray(log_probs, dtype=np.float32)
    B, T, U1, V = lp.shape
    y = np.ascontiguousarray(labels, dtype=np.int32).reshape(B, max(U1 - 1, 0))
    tl = np.ascontiguousarray(frames_lengths, dtype=np.int32)
    ul = np.ascontiguousarray(labels_lengths, dtype=np.int32)
    dt = np.dtype(dtype)
    fn = lib().oracle_rnnt_loss_f64 if dt == np.float64 else lib().oracle_rnnt_loss_f32
    costs = np.empty(B, dt)
    grads = np.empty((B, T, U1, V), dt) if want_grads else None
    alphas = np.empty((B, T, U1), dt) if want_lattice else None
    betas = np.empty((B, T, U1), dt) if want_lattice else None
    rc = fn(_ptr(lp), _ptr(y), _ptr(tl), _ptr(ul), B, T, U1, V, blank,
            _ptr(costs), _ptr(grads), _ptr(alphas), _ptr(betas))
    if rc != 0:
        raise ValueError("oracle_rnnt_loss: length out of range")
    if want_lattice:
        return costs, grads, alphas, betas
    return costs, grads


def brute_force_cost(lp, y, blank=0):
    """-log sum over every alignment path; pure Python, tiny lattices only.

    lp: (T,U1,V) float64 log-probs of ONE utterance with T_n=T, U_n=U1-1; y: (U1-1,).
    A path starts at (0,0), each step either emits blank (t+=1) or y[u] (u+=1),
    and ends with a blank from (T-1,U).
    """
    T, U1, _ = lp.shape
    U = U1 - 1
    total = []

    def walk(t, u, acc):
        if t == T - 1 and u == U:
            total.append(acc + lp[t, u, blank])
            return
        if t < T - 1:
            walk(t + 1, u, acc + lp[t, u, blank])
        if u < U:
            walk(t, u + 1, acc + lp[t, u, y[u]])

    walk(0, 0, 0.0)
    return -float(np.logaddexp.reduce(np.array(total, dtype=np.float64)))
