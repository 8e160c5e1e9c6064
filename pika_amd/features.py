"""Feature-side ops of the training step on the GPU: CMVN and SpecAugment.

`SpecAugment` mirrors /root/reference/utils/spec_augment.py:3-20 exactly on the host side --
the same two draws from torch's CPU RNG (`Uniform.sample`) and the same two draws from numpy's
global RNG, in the same order -- so that with identical seeds the same bands are masked; the
masking itself is one HIP kernel instead of two strided slice-assign kernels.
"""
import numpy as np
import torch
from torch.distributions.uniform import Uniform

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_hip(x, what):
    if not x.is_cuda:
        raise RuntimeError("pika_amd.%s: tensor must live on a HIP device (no CPU path)" % what)
    if x.dtype != torch.float32 or x.dim() != 3 or not x.is_contiguous():
        raise ValueError("pika_amd.%s: expected a contiguous float32 (B,T,F) tensor" % what)


def specaug_apply_(x, f0, fs, t0, ts):
    """Zero x[:, :, f0:f0+fs] and x[:, t0:t0+ts, :] in place (one launch)."""
    _require_hip(x, "specaug_apply_")
    B, T, F = x.shape
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().pika_specaug_apply(x.data_ptr(), B, T, F, int(f0), int(fs), int(t0),
                                                 int(ts), _stream()), "pika_specaug_apply")
    return x


def cmvn_apply_(x, offset=None, scale=None, cmn=True):
    """In place: optional per-utterance mean removal over time, then (x + offset) * scale
    (train_transducer_bmuf_otfaug.py:86-91)."""
    _require_hip(x, "cmvn_apply_")
    B, T, F = x.shape
    for v in (offset, scale):
        if v is not None and (v.dtype != torch.float32 or v.numel() != F or v.device != x.device):
            raise ValueError("offset/scale must be float32 (F,) on the same device")
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().pika_cmvn_apply(
            x.data_ptr(), B, T, F, None if offset is None else offset.contiguous().data_ptr(),
            None if scale is None else scale.contiguous().data_ptr(), int(bool(cmn)), _stream()),
            "pika_cmvn_apply")
    return x


class SpecAugment(object):
    """Drop-in for utils.spec_augment.SpecAugment (same ctor, `.apply(inp)` in place)."""

    def __init__(self, max_freq_span, max_time_span, batch_first=True):
        self.freq_span_sampler = Uniform(0.0, float(max_freq_span + 1))
        self.time_span_sampler = Uniform(0.0, float(max_time_span + 1))
        self.batch_first = batch_first

    def draw(self, T, F):
        """The reference's RNG protocol (spec_augment.py:13-19): returns (f0, fs, t0, ts)."""
        freq_span = int(self.freq_span_sampler.sample().item())
        time_span = int(self.time_span_sampler.sample().item())
        f0 = t0 = 0
        if freq_span > 0:
            f0 = np.random.randint(0, F - freq_span)
        if time_span > 0:
            t0 = np.random.randint(0, T - time_span)
        return f0, freq_span, t0, time_span

    def apply(self, inp):
        f0, fs, t0, ts = self.draw(inp.size()[1], inp.size()[-1])
        if fs > 0 or ts > 0:
            specaug_apply_(inp, f0, fs, t0, ts)
