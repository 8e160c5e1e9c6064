#!/usr/bin/env python
"""Which n-best order is RIGHT where the GPU's and the fp32 reference's differ?  Both lists against the float64 run of the
reference decoder (tests/golden/decode_full_f64.npz, make_decode_full_f64_golden.py): entries present in the float64
finished list, entries at their float64 rank, score error against float64.      python tools/decode_f64_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def against_f64(lst, f, b, nb):
    """(found, at_rank, max |score - s64|) of list `lst` for utterance b; an entry matches the float64 entry with the same
    symbols whose score is nearest (the reference's finished list can hold one symbol sequence twice)."""
    found = at_rank = 0
    err = 0.0
    for j in range(nb):
        L = int(lst["lens"][b, j])
        ks = [k for k in range(int(f["count"][b])) if int(f["lens"][b, k]) == L and
              np.array_equal(f["hyps"][b, k, :L], lst["hyps"][b, j, :L])]
        if not ks:
            continue
        k = min(ks, key=lambda k_: abs(f["scores"][b, k_] - lst["scores"][b, j]))
        found += 1
        at_rank += int(k == j)
        err = max(err, abs(float(f["scores"][b, k]) - float(lst["scores"][b, j])))
    return found, at_rank, err


if __name__ == "__main__":
    import torch
    import test_decode_full as T
    z = np.load(T.GOLD)
    f = np.load(os.path.join(ROOT, "tests", "golden", "decode_full_f64.npz"))
    dev = torch.device("cuda:0")
    B, nb = z["lens"].shape
    for mode in ("fp32", "fp32-exact"):
        got, _, _ = T.decode(dev, mode)
        for b in range(B):
            print(mode, "utt", b, "GPU (found, at f64 rank, max err):", against_f64(got, f, b, nb),
                  "| fp32 reference:", against_f64(z, f, b, nb))
