"""Generates tests/golden/specaug.npz by running the REFERENCE utils/spec_augment.py (CPU):
for each seed, seed torch + numpy, apply to a (3,120,240) tensor of ones, record the zero mask
bands.  python tests/golden/make_specaug_golden.py   (needs /root/reference)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from utils.spec_augment import SpecAugment  # the reference class

HERE = os.path.dirname(os.path.abspath(__file__))
B, T, F = 3, 120, 240
seeds = list(range(40))
bands = []
for s in seeds:
    torch.manual_seed(s)
    np.random.seed(s)
    aug = SpecAugment(15, 35)
    for rep in range(2):  # two consecutive batches per seed: RNG streams must stay in step
        x = torch.ones(B, T, F)
        aug.apply(x)
        z = (x[0] == 0).numpy()
        full_rows = np.where(z.all(axis=1))[0]
        t0, ts = (int(full_rows[0]), len(full_rows)) if len(full_rows) else (0, 0)
        keep = np.ones(T, bool)
        keep[full_rows] = False
        cols = np.where(z[keep].all(axis=0))[0] if keep.any() else np.array([], int)
        f0, fs = (int(cols[0]), len(cols)) if len(cols) else (0, 0)
        assert (x[1] == x[0]).all() and (x[2] == x[0]).all()
        bands.append((s, rep, f0, fs, t0, ts))
np.savez(os.path.join(HERE, "specaug.npz"), bands=np.array(bands, np.int64), shape=np.array([B, T, F]))
print("wrote specaug.npz;", sum(1 for b in bands if b[3] == 0), "draws without a freq band,",
      sum(1 for b in bands if b[5] == 0), "without a time band")
