"""Golden for pika_amd/loader/augment.py: runs the REFERENCE AudioSegment.add_noise / convolve_and_normalize
(loader/audio.py, imported from /root/reference with stub soundfile/resampy modules) on seeded signals.
    python tests/golden/make_augment_golden.py"""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for name in ("soundfile", "resampy"):
    sys.modules.setdefault(name, types.ModuleType(name))
if not hasattr(np, "sctypes"):
    np.sctypes = {"int": [np.int8, np.int16, np.int32, np.int64], "float": [np.float16, np.float32, np.float64]}
sys.path.insert(0, "/root/reference")
from loader.audio import AudioSegment  # noqa: E402

rng = np.random.default_rng(21)
out = {}
k = 0
for n, m, snr, seed in [(16000, 40000, 10.0, 1), (12345, 12345, 0.0, 2), (8000, 100000, 25.5, 3)]:
    sig = (rng.standard_normal(n) * 0.1).astype(np.float32)
    noi = (rng.standard_normal(m) * 0.03).astype(np.float32)
    a, b = AudioSegment(sig.copy(), 16000), AudioSegment(noi.copy(), 16000)
    try:
        a.add_noise(b, snr, rng=random.Random(seed))
        res, err = a.samples.astype(np.float32), 0
    except ValueError:
        res, err = np.zeros(0, np.float32), 1
    out["n%d/sig" % k], out["n%d/noise" % k], out["n%d/out" % k] = sig, noi, res
    out["n%d/cfg" % k] = np.array([snr, seed, err])
    k += 1
out["n_noise"] = np.array(k)
k = 0
for n, m in [(16000, 1), (20000, 801), (9000, 4000), (5000, 2500)]:
    sig = (rng.standard_normal(n) * 0.1).astype(np.float32)
    h = (rng.standard_normal(m) * np.exp(-np.arange(m) / max(m / 6.0, 1.0))).astype(np.float32)
    a = AudioSegment(sig.copy(), 16000)
    a.convolve_and_normalize(AudioSegment(h.copy(), 16000))
    out["c%d/sig" % k], out["c%d/rir" % k], out["c%d/out" % k] = sig, h, a.samples.astype(np.float32)
    k += 1
out["n_conv"] = np.array(k)
np.savez_compressed(os.path.join(HERE, "augment.npz"), **out)
print("wrote augment.npz", {kk: v.shape for kk, v in out.items() if kk.endswith("/out")})
