"""oracle/fbank_ref.py -- TEST INFRASTRUCTURE: numpy restatement of the feature pipeline of
loader/otf_utt_loader.py:213-250.

* `kaldi_fbank`: Kaldi's `Fbank.compute_features` (reference call site
  loader/otf_utt_loader.py:195-201,231-234; options egs/fbank.conf:1-6).  The arithmetic lives in
  third-party Kaldi reached through PyKaldi (README.md:30-32, version unpinned, absent from this
  container and from /root/reference): PARITY UNPINNED -- no reference test pins it.  Restated
  from Kaldi's published feature-window / feature-fbank / mel-computations algorithm:
  snip-edges framing (25 ms / 10 ms), optional dither, DC removal, pre-emphasis 0.97, Hamming
  window, zero-pad to 512, power spectrum, 80 triangular mel bins on [40, 7800] Hz over FFT bins
  0..255, floor at FLT_EPSILON, log.  float64 internally (truth); dither=0 for parity runs
  (the recipe's dither=1 makes the reference itself non-deterministic).  Cross-checked (tests/
  test_fbank_crosscheck.py) against an independent published implementation of the same Kaldi algorithm,
  transformers.audio_utils (mel_scale="kaldi", triangularize_in_mel_space): agreement to 1e-7 on the log-mel
  values for noise / tone / chirp / near-silent / one-frame inputs and two bank layouts.  That is a second
  witness, not the reference's own PyKaldi build, so the row stays formally unpinned.
* `change_speed` / `normalize` / `to_int16`: restated from loader/audio.py:217-262,551-603 and
  cross-checked against the reference class itself in tests/golden/make_audio_golden.py.
* `splice`: loader/otf_utt_loader.py:28-46.
"""
import numpy as np


def mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_banks(num_bins=80, sample_freq=16000.0, nfft=512, low=40.0, high=-200.0):
    nyq = 0.5 * sample_freq
    if high <= 0:
        high += nyq
    nb = nfft // 2  # Kaldi ignores the Nyquist bin
    bw = sample_freq / nfft
    ml, mh = mel(low), mel(high)
    delta = (mh - ml) / (num_bins + 1)
    w = np.zeros((num_bins, nb), np.float64)
    for b in range(num_bins):
        left, center, right = ml + b * delta, ml + (b + 1) * delta, ml + (b + 2) * delta
        for i in range(nb):
            m = mel(bw * i)
            if left < m < right:
                w[b, i] = (m - left) / (center - left) if m <= center else (right - m) / (right - center)
    return w


def kaldi_fbank(wave, sample_freq=16000.0, num_bins=80, low=40.0, high=-200.0, dither=0.0,
                preemph=0.97, frame_len_ms=25.0, frame_shift_ms=10.0, rng=None):
    wave = np.asarray(wave, np.float64)
    flen = int(sample_freq * 0.001 * frame_len_ms)
    shift = int(sample_freq * 0.001 * frame_shift_ms)
    nfft = 1
    while nfft < flen:
        nfft *= 2
    if len(wave) < flen:
        return np.zeros((0, num_bins), np.float64)
    n = 1 + (len(wave) - flen) // shift
    window = 0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(flen) / (flen - 1))
    banks = mel_banks(num_bins, sample_freq, nfft, low, high)
    out = np.zeros((n, num_bins), np.float64)
    for f in range(n):
        w = wave[f * shift: f * shift + flen].copy()
        if dither != 0.0:
            w += dither * (rng or np.random).standard_normal(flen)
        w -= w.mean()
        w[1:] -= preemph * w[:-1]
        w[0] -= preemph * w[0]
        w *= window
        spec = np.fft.rfft(w, nfft)
        power = (spec.real ** 2 + spec.imag ** 2)[:nfft // 2]
        e = banks @ power
        out[f] = np.log(np.maximum(e, np.finfo(np.float32).eps))
    return out


def change_speed(samples_f32, rate):
    """loader/audio.py:217-238 (np.interp on linspace(0, N, int(N/rate)))."""
    if rate == 1.0:
        return samples_f32
    n = samples_f32.shape[0]
    m = int(n / rate)
    return np.interp(np.linspace(0, n, m), np.arange(n), samples_f32)


def normalize(samples, target_db, max_gain_db=300.0):
    """loader/audio.py:240-262,207-215,551-560."""
    rms_db = 10 * np.log10(max(1e-20, np.mean(samples ** 2)))
    return samples * 10.0 ** (min(max_gain_db, target_db - rms_db) / 20.0)


def to_int16(samples):
    """loader/audio.py:578-603: scale by 2^15, clip, astype(int16) (truncation)."""
    out = samples * 32768.0
    return np.clip(out, -32768, 32767).astype(np.int16)


def perturb(pcm_i16, rate, target_db):
    """int16 PCM -> int16 PCM exactly as loader/otf_utt_loader.py:218-230."""
    x = pcm_i16.astype(np.float32) * np.float32(1.0 / 32768)
    return to_int16(normalize(change_speed(x, rate), target_db))


def splice(feats, lctx, rctx):
    n, d = feats.shape
    pad = np.concatenate([np.repeat(feats[:1], lctx, 0), feats, np.repeat(feats[-1:], rctx, 0)])
    return np.concatenate([pad[i:i + n] for i in range(lctx + 1 + rctx)], axis=1).astype(np.float32)
