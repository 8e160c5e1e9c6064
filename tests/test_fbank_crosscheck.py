"""The fbank oracle (oracle/fbank_ref.py, a restatement of Kaldi's feature-fbank written for this repo) against an
INDEPENDENT published implementation of the same Kaldi algorithm: `transformers.audio_utils` (offline wheel in this
image; `mel_scale="kaldi"`, `triangularize_in_mel_space=True`, `remove_dc_offset`, `preemphasis` are the options its
Kaldi-compatible feature extractors use, and upstream tests those against torchaudio.compliance.kaldi).  Neither is the
PyKaldi build the reference calls (loader/otf_utt_loader.py:195-201,231-234; absent here), so SURVEY 8c still lists the
row as unpinned -- but the oracle the HIP kernel is tested against (tests/test_features.py) is no longer its only witness.
dither = 0 on both sides (egs/fbank.conf's dither=1 is noise by definition)."""
import numpy as np
import pytest

from oracle.fbank_ref import kaldi_fbank

A = pytest.importorskip("transformers.audio_utils")


def third_party_fbank(wave, num_bins=80, low=40.0, high=7800.0, flen=400, shift=160, nfft=512):
    win = A.window_function(flen, "hamming", periodic=False)
    mel = A.mel_filter_bank(num_frequency_bins=nfft // 2 + 1, num_mel_filters=num_bins, min_frequency=low,
                            max_frequency=high, sampling_rate=16000, norm=None, mel_scale="kaldi",
                            triangularize_in_mel_space=True)
    return A.spectrogram(np.asarray(wave, np.float64), win, frame_length=flen, hop_length=shift, fft_length=nfft,
                         power=2.0, center=False, preemphasis=0.97, mel_filters=mel,
                         mel_floor=float(np.finfo(np.float32).eps), log_mel="log", remove_dc_offset=True,
                         dtype=np.float64).T


def waves():
    rng = np.random.default_rng(0)
    noise = (rng.standard_normal(32000) * 3000).clip(-32768, 32767).round()
    tone = noise.copy()
    tone[8000:12000] += 8000 * np.sin(2 * np.pi * 440 * np.arange(4000) / 16000)
    quiet = (rng.standard_normal(5000) * 2).round()                    # a few LSBs: exercises the energy floor region
    chirp = 12000 * np.sin(2 * np.pi * (50 + 3900 * np.linspace(0, 1, 24000)) * np.arange(24000) / 16000)
    short = noise[:400]                                                # exactly one frame
    return {"noise": noise, "tone": tone, "quiet": quiet, "chirp": chirp.round(), "one_frame": short}


@pytest.mark.parametrize("name", sorted(waves()))
def test_oracle_equals_the_third_party_kaldi_fbank(name):
    w = waves()[name]
    ours = kaldi_fbank(w, dither=0.0)
    theirs = third_party_fbank(w)
    assert ours.shape == theirs.shape == (1 + (len(w) - 400) // 160, 80)
    # their spectrum passes through complex64: 1e-6 relative on the power spectrum -> 1e-6 absolute on the log
    assert np.abs(ours - theirs).max() < 5e-6, np.abs(ours - theirs).max()


def test_other_bank_layouts_agree_too():
    """40 bins on [20, 8000] Hz (high-freq 0 = Nyquist): the mel-bank construction, not just one table."""
    w = waves()["tone"]
    ours = kaldi_fbank(w, num_bins=40, low=20.0, high=0.0, dither=0.0)
    theirs = third_party_fbank(w, num_bins=40, low=20.0, high=8000.0)
    assert np.abs(ours - theirs).max() < 5e-6
