"""Shared seeded-input builders for the parity tests (numpy only; no product imports)."""
import numpy as np


def log_softmax(x):
    x = x - x.max(-1, keepdims=True)
    return x - np.log(np.exp(x).sum(-1, keepdims=True))


def make_case(B, T, U, V, seed, ragged=False, scale=1.0, blank=0):
    """log_probs f32 (B,T,U+1,V), labels i32 (B,U) padded with V, lengths i32."""
    rng = np.random.default_rng(seed)
    lp = log_softmax(rng.standard_normal((B, T, U + 1, V)) * scale).astype(np.float32)
    lo = 1 if blank == 0 else 0
    labels = rng.integers(lo, V, (B, U)).astype(np.int32)
    if blank != 0:
        labels[labels == blank] = (blank + 1) % V
    if ragged:
        tl = rng.integers(max(1, T // 2), T + 1, B).astype(np.int32)
        ul = rng.integers(0, U + 1, B).astype(np.int32)
        tl[0], ul[0] = T, U  # keep one full-size utterance
        if B > 1:
            tl[1], ul[1] = 1, 0  # the smallest legal lattice
    else:
        tl = np.full(B, T, np.int32)
        ul = np.full(B, U, np.int32)
    for n in range(B):
        labels[n, ul[n]:] = V  # reference padding value (egs/train_transducer_bmuf_otfaug.sh:38)
    return lp, labels, tl, ul
