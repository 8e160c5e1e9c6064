"""Loader front end.  CPU: the oracle restatements vs golden from the
reference AudioSegment; the product's mel plan vs the oracle's.  GPU: HIP kernels vs oracle."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
from oracle import fbank_ref as F  # noqa: E402

GOLD = os.path.join(HERE, "golden", "audio_perturb.npz")


def test_oracle_perturb_matches_reference_golden():
    z = np.load(GOLD)
    for i in range(int(z["n"])):
        rate, db = z["cfg%d" % i]
        assert np.array_equal(F.perturb(z["pcm%d" % i], float(rate), float(db)), z["out%d" % i])


def test_oracle_fbank_sanity():
    # a pure tone lands in the mel bin that covers its frequency; shapes follow snip-edges
    sr, f0 = 16000, 1000.0
    t = np.arange(sr) / sr
    wave = 8000 * np.sin(2 * np.pi * f0 * t)
    fb = F.kaldi_fbank(wave)
    assert fb.shape == (98, 80)
    banks = F.mel_banks()
    assert fb.mean(0).argmax() == banks[:, int(round(f0 / (sr / 512)))].argmax()
    assert np.allclose(banks.sum(0)[3:240], 1.0, atol=1e-9)  # triangles partition unity inside the band


def test_product_mel_plan_equals_oracle():
    from pika_amd.loader.frontend import FbankConfig
    cfg = FbankConfig(num_mel_bins=80, low_freq=40, high_freq=-200, dither=0.0, window_type="hamming")
    lo, cnt, ptr, w = cfg.mel_plan()
    dense = np.zeros((80, 256))
    for b in range(80):
        dense[b, lo[b]:lo[b] + cnt[b]] = w[ptr[b]:ptr[b] + cnt[b]]
    assert np.allclose(dense, F.mel_banks(), atol=1e-6)
    assert cfg.frame_len == 400 and cfg.shift == 160 and cfg.nfft == 512


def test_fbank_conf_parser(tmp_path):
    from pika_amd.loader.frontend import FbankConfig
    p = tmp_path / "fbank.conf"
    p.write_text("--window-type=hamming \n--sample-frequency=16000\n--dither=1\n--low-freq=40    # c\n"
                 "--high-freq=-200 # c\n--num-mel-bins=80\n")
    cfg = FbankConfig.from_file(str(p))
    assert (cfg.num_mel_bins, cfg.low_freq, cfg.high_freq, cfg.dither) == (80, 40.0, -200.0, 1.0)


@pytest.mark.gpu
def test_gpu_perturb_matches_reference_golden(hip_device):
    from pika_amd.loader.frontend import FbankConfig, GpuFrontEnd
    z = np.load(GOLD)
    n = int(z["n"])
    cfg = FbankConfig(num_mel_bins=80, low_freq=40, high_freq=-200, dither=0.0, window_type="hamming")
    fe = GpuFrontEnd(cfg, hip_device)
    pcms = [z["pcm%d" % i] for i in range(n)]
    rates = [float(z["cfg%d" % i][0]) for i in range(n)]
    dbs = [float(z["cfg%d" % i][1]) for i in range(n)]
    fe(pcms, rates, dbs)
    wave = fe.last_wave.cpu().numpy()
    off = fe.last_offsets[1]
    for i in range(n):
        got = wave[off[i]:off[i + 1]]
        want = z["out%d" % i].astype(np.float32)
        assert got.shape == want.shape
        diff = np.abs(got - want)
        # fp64 on both sides; only the fp32 unchanged-speed branch may differ by one LSB, rarely
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (i, diff.max(), (diff > 0).mean())


@pytest.mark.gpu
def test_gpu_fbank_splice_match_oracle(hip_device):
    from pika_amd.loader.frontend import FbankConfig, GpuFrontEnd
    rng = np.random.default_rng(5)
    pcms = [np.clip(rng.standard_normal(n) * 2500, -32768, 32767).astype(np.int16)
            for n in (16000, 5000, 400, 12345)]
    t = np.arange(8000) / 16000.0
    pcms.append((6000 * np.sin(2 * np.pi * 440 * t) + 3000 * np.sin(2 * np.pi * 3000 * t)).astype(np.int16))
    cfg = FbankConfig(num_mel_bins=80, low_freq=40, high_freq=-200, dither=0.0, window_type="hamming")
    fe = GpuFrontEnd(cfg, hip_device, lctx=1, rctx=1, stride=1)
    data, lens = fe(pcms, [1.0] * 5, [0.0] * 5, perturb=False)
    data = data.cpu().numpy()
    assert lens == [98, 29, 1, 75, 48]
    for b, pcm in enumerate(pcms):
        ref = F.kaldi_fbank(pcm.astype(np.float64))
        sp = F.splice(ref.astype(np.float32), 1, 1)
        got = data[b, :lens[b]]
        # log-mel in fp32 (FFT + mel sum): 2e-4 absolute on values O(10)
        assert np.abs(got - sp).max() < 2e-3, (b, np.abs(got - sp).max())
        if lens[b] < data.shape[1]:
            assert np.array_equal(data[b, lens[b]:], np.repeat(got[-1:], data.shape[1] - lens[b], 0))
    # stride 3 + wider context
    fe2 = GpuFrontEnd(cfg, hip_device, lctx=2, rctx=1, stride=3)
    d2, l2 = fe2(pcms[:2], [1.0, 1.0], [0.0, 0.0], perturb=False)
    for b in range(2):
        ref = F.splice(F.kaldi_fbank(pcms[b].astype(np.float64)).astype(np.float32), 2, 1)[::3]
        assert l2[b] == ref.shape[0] and np.abs(d2[b, :l2[b]].cpu().numpy() - ref).max() < 2e-3


@pytest.mark.gpu
def test_gpu_dither_is_statistical(hip_device):
    from pika_amd.loader.frontend import FbankConfig, GpuFrontEnd
    cfg = FbankConfig(num_mel_bins=80, low_freq=40, high_freq=-200, dither=1.0, window_type="hamming")
    fe = GpuFrontEnd(cfg, hip_device)
    pcm = np.zeros(16000, np.int16)  # silence: the output is the spectrum of the dither alone
    a, _ = fe([pcm], [1.0], [0.0], perturb=False)
    b, _ = fe([pcm], [1.0], [0.0], perturb=False)
    assert not torch.equal(a, b)                      # fresh noise per call
    rng = np.random.default_rng(0)
    ref = F.kaldi_fbank(np.zeros(16000), dither=1.0, rng=rng)
    got = a[0, :, 80:160].cpu().numpy()
    assert abs(got.mean() - ref.mean()) < 0.15 and abs(got.std() - ref.std()) < 0.15


@pytest.mark.gpu
def test_noise_and_reverb_augmentation_match_reference(hip_device):
    """pika_amd/loader/augment.py vs golden from the reference AudioSegment.add_noise /
    convolve_and_normalize (tests/golden/make_augment_golden.py): same RNG draw and index rounding, same gains."""
    import random
    from pika_amd.loader import augment as A
    z = np.load(os.path.join(HERE, "golden", "augment.npz"))
    for k in range(int(z["n_noise"])):
        snr, seed, err = z["n%d/cfg" % k]
        sig = torch.from_numpy(z["n%d/sig" % k]).to(hip_device)
        noi = torch.from_numpy(z["n%d/noise" % k]).to(hip_device)
        if err:
            with pytest.raises(ValueError):
                A.add_noise_(sig, noi, float(snr), rng=random.Random(int(seed)))
            continue
        A.add_noise_(sig, noi, float(snr), rng=random.Random(int(seed)))
        want = z["n%d/out" % k]
        assert np.abs(sig.cpu().numpy() - want).max() < 2e-6 * max(1.0, np.abs(want).max())
    for k in range(int(z["n_conv"])):
        sig = torch.from_numpy(z["c%d/sig" % k]).to(hip_device)
        rir = torch.from_numpy(z["c%d/rir" % k]).to(hip_device)
        got = A.convolve_and_normalize(sig, rir).cpu().numpy()
        want = z["c%d/out" % k]
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 1e-4 * np.abs(want).max(), k
    with pytest.raises(RuntimeError):
        A.rms_db(torch.zeros(4))                   # no CPU path


@pytest.mark.gpu
def test_side_stream_prefetch_gives_the_synchronous_batches(hip_device):
    """The loader's device half on its own stream, driven two batches ahead by DevicePrefetcher (pinned ring, one
    upload per batch, event hand-off to the consumer's stream), returns exactly what the synchronous front end
    returns -- including when the consumer's stream is busy and the ring wraps around -- and a new front end (next
    epoch / other rank) draws different dither noise."""
    import queue
    from types import SimpleNamespace
    from pika_amd.loader.frontend import FbankConfig, GpuFrontEnd
    from pika_amd.loader import otf_utt_loader as L
    cfg = FbankConfig(num_mel_bins=80, low_freq=40, high_freq=-200, dither=0.0, window_type="hamming")
    rng = np.random.default_rng(21)
    raw_batches = []
    for n_b in range(7):                                   # > 2 x ring slots: the ring wraps
        batch = []
        for _ in range(3):
            n = int(rng.integers(3000, 20000))
            pcm = np.clip(rng.standard_normal(n) * 2500, -32768, 32767).astype(np.int16)
            rate = [0.9, 1.0, 1.1][int(rng.integers(0, 3))]
            n_out = n if rate == 1.0 else int(n / rate)
            batch.append((pcm, rate, float(rng.uniform(-50, -10)), rng.integers(1, 50, 4).astype(np.int32),
                          cfg.num_frames(n_out)))
        raw_batches.append(batch)
    args = SimpleNamespace(padding_tgt=99, batch_first=True)
    sync_fe = GpuFrontEnd(cfg, hip_device, 1, 1, 1)
    want = [L.assemble(b, sync_fe, args) for b in raw_batches]
    q = queue.Queue()
    for b in raw_batches:
        q.put(b)
    q.put(None)
    fe = GpuFrontEnd(cfg, hip_device, 1, 1, 1, side_stream=True)
    big = torch.randn(4096, 4096, device=hip_device)
    got = []
    for batch in L.DevicePrefetcher(q, 1, fe, args):
        for _ in range(3):
            big = big @ big * 1e-3                          # keep the consumer's stream busy
        got.append([t.clone() if torch.is_tensor(t) else t for t in batch])
    assert len(got) == len(want) and fe.batches == 7 and fe.host_seconds > 0
    for g, w in zip(got, want):
        # with a side stream the targets and lengths arrive on the device with the batch (one pinned upload on the front
        # end's stream); the synchronous front end hands back host tensors as before
        assert all(t.is_cuda for t in g) and not any(t.is_cuda for t in w[1:])
        assert torch.equal(g[0], w[0]) and all(torch.equal(g[i].cpu(), w[i].to(torch.int32)) for i in (1, 2, 3))
    # dither: (base seed, instance, batch) keyed -- a second front end never replays the first one's noise
    cfg_d = FbankConfig(num_mel_bins=80, low_freq=40, high_freq=-200, dither=1.0, window_type="hamming")
    pcm = np.zeros(16000, np.int16)
    a = GpuFrontEnd(cfg_d, hip_device, base_seed=7)([pcm], [1.0], [0.0], perturb=False)[0]
    b = GpuFrontEnd(cfg_d, hip_device, base_seed=7)([pcm], [1.0], [0.0], perturb=False)[0]
    assert not torch.equal(a, b)
