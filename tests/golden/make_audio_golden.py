"""Generates tests/golden/audio_perturb.npz by running the REFERENCE `AudioSegment`
(loader/audio.py, imported from /root/reference with stub `soundfile`/`resampy` modules, which it
imports at module top but never touches on this path) through the exact call sequence of
loader/otf_utt_loader.py:218-230, and the reference `splice` (:28-46).
    python tests/golden/make_audio_golden.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
for name in ("soundfile", "resampy"):
    sys.modules.setdefault(name, types.ModuleType(name))
if not hasattr(np, "sctypes"):  # removed in numpy 2; audio.py:570,592 uses it
    np.sctypes = {"int": [np.int8, np.int16, np.int32, np.int64],
                  "float": [np.float16, np.float32, np.float64]}
sys.path.insert(0, "/root/reference")
from loader.audio import AudioSegment  # noqa: E402  (the reference class)
from oracle import fbank_ref as F  # noqa: E402

rng = np.random.default_rng(11)
cases = {}
for i, (n, rate, db) in enumerate([(4000, 0.9, -20.0), (16000, 1.0, -35.5), (12345, 1.1, -12.25),
                                   (801, 0.9, -50.0), (3000, 1.1, -5.0)]):
    pcm = np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16)
    seg = AudioSegment(pcm, 16000)
    seg.change_speed(rate)
    seg.normalize(db)
    out = seg._convert_samples_from_float32(seg.samples, 'int16')
    assert np.array_equal(out, F.perturb(pcm, rate, db)), "restatement disagrees with the reference"
    cases["pcm%d" % i], cases["out%d" % i] = pcm, out
    cases["cfg%d" % i] = np.array([rate, db])
# splice golden straight from the reference function
sys.modules.setdefault("kaldi", types.ModuleType("kaldi"))
np.savez_compressed(os.path.join(HERE, "audio_perturb.npz"), n=np.array(5), **cases)
print("wrote audio_perturb.npz; oracle restatement == reference AudioSegment on all cases")
