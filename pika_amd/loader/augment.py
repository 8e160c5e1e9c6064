"""Noise and reverberation augmentation on the GPU: the arithmetic of the reference's
AudioSegment.add_noise / convolve_and_normalize (loader/audio.py:426-513) on device float tensors, with the
reference's RNG draw (`rng.uniform(0, noise_duration - duration)`) and its sample-index rounding.

The hooks that would call these are commented out in the reference loader (otf_utt_loader.py:224-228, rir / noise
lists plumbed but empty), so nothing in the RNN-T recipes depends on them yet."""
import math
import random

import torch

from .. import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_hip(t):
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("pika_amd.loader.augment: contiguous float32 HIP tensors only (no CPU path)")


def rms_db(x):
    """10 log10(max(1e-20, mean(x^2)))  (audio.py:552-560); one device reduction, one host read."""
    _need_hip(x)
    acc = torch.empty(1, dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().pika_audio_sumsq(x.data_ptr(), x.numel(), acc.data_ptr(), _stream()), "pika_audio_sumsq")
    return 10.0 * math.log10(max(1e-20, float(acc.item()) / x.numel()))


def gain_db_(x, gain):
    """x *= 10^(gain/20) in place (audio.py:207-215)."""
    _need_hip(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().pika_audio_axpby(x.data_ptr(), None, x.numel(), 0.0, 10.0 ** (gain / 20.0), _stream()),
                   "pika_audio_axpby")
    return x


def add_noise_(signal, noise, snr_db, sample_rate=16000, max_gain_db=300.0, rng=None):
    """In place: signal += gain * (random subsegment of noise), gain from the SNR (audio.py:467-513).
    Raises ValueError where the reference does (noise shorter than the signal; subsegment length off by one
    sample after the reference's rounding of start / end times)."""
    _need_hip(signal); _need_hip(noise)
    rng = random.Random() if rng is None else rng
    n, m = signal.numel(), noise.numel()
    dur, ndur = n / float(sample_rate), m / float(sample_rate)
    if ndur < dur:
        raise ValueError("Noise signal (%f sec) must be at least as long as base signal (%f sec)." % (ndur, dur))
    noise_gain_db = min(rms_db(signal) - rms_db(noise) - snr_db, max_gain_db)
    start = rng.uniform(0.0, ndur - dur)                                   # random_subsegment :411-424
    s0, s1 = int(round(start * sample_rate)), int(round((start + dur) * sample_rate))   # subsegment :403-405
    seg = noise[s0:s1]
    if seg.numel() != n:
        raise ValueError("Segment lengths must match to add segments.")    # superimpose :191-192
    with torch.cuda.device(signal.device):
        _lib.check(_lib.lib().pika_audio_axpby(signal.data_ptr(), seg.data_ptr(), n, 10.0 ** (noise_gain_db / 20.0),
                                               1.0, _stream()), "pika_audio_axpby")
    return signal


def convolve_and_normalize(signal, impulse, max_gain_db=300.0):
    """fftconvolve(signal, impulse, "same") re-normalised to the input's RMS level (audio.py:426-465, 240-262).
    Returns a new tensor."""
    _need_hip(signal); _need_hip(impulse)
    if impulse.numel() > signal.numel():
        raise ValueError("impulse response longer than the signal is not supported")
    target_db = rms_db(signal)
    out = torch.empty_like(signal)
    with torch.cuda.device(signal.device):
        _lib.check(_lib.lib().pika_audio_convolve_same(signal.data_ptr(), signal.numel(), impulse.data_ptr(),
                                                       impulse.numel(), out.data_ptr(), _stream()),
                   "pika_audio_convolve_same")
    gain = target_db - rms_db(out)
    if gain > max_gain_db:
        raise ValueError("Unable to normalize segment to %f dB because the the probable gain have exceeds "
                         "max_gain_db (%f dB)" % (target_db, max_gain_db))
    return gain_db_(out, min(max_gain_db, gain))
