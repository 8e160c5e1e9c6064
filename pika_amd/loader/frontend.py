"""GPU feature front end: int16 PCM of a whole batch -> (B, Tmax, feats_dim*(lctx+1+rctx)) f32.

Mirrors what loader/otf_utt_loader.py:213-270 does per utterance on the CPU (AudioSegment
perturbation -> Kaldi fbank -> splice -> pad), as three HIP kernels over the concatenated batch
(include/pika_audio.h), fed from ONE pinned host buffer per batch.
"""
import math

import numpy as np
import torch

from .. import _lib


class FbankConfig(object):
    """The options of egs/fbank.conf (Kaldi `--key=value` lines); everything else = Kaldi defaults."""

    def __init__(self, sample_frequency=16000.0, num_mel_bins=23, low_freq=20.0, high_freq=0.0,
                 dither=1.0, preemphasis_coefficient=0.97, frame_length=25.0, frame_shift=10.0,
                 window_type="povey"):
        self.sample_frequency = float(sample_frequency)
        self.num_mel_bins = int(num_mel_bins)
        self.low_freq, self.high_freq = float(low_freq), float(high_freq)
        self.dither = float(dither)
        self.preemphasis_coefficient = float(preemphasis_coefficient)
        self.frame_length, self.frame_shift = float(frame_length), float(frame_shift)
        self.window_type = window_type

    @classmethod
    def from_file(cls, path):
        kw = {}
        with open(path) as f:
            for line in f:
                line = line.split("#")[0].strip()
                if not line.startswith("--") or "=" not in line:
                    continue
                k, v = line[2:].split("=", 1)
                k = k.strip().replace("-", "_")
                v = v.strip()
                kw[k] = v if k == "window_type" else float(v)
        cfg = cls(**kw)
        if cfg.window_type != "hamming":
            raise NotImplementedError("only --window-type=hamming (egs/fbank.conf:1) is implemented")
        return cfg

    @property
    def frame_len(self):
        return int(self.sample_frequency * 0.001 * self.frame_length)

    @property
    def shift(self):
        return int(self.sample_frequency * 0.001 * self.frame_shift)

    @property
    def nfft(self):
        n = 1
        while n < self.frame_len:
            n *= 2
        return n

    def num_frames(self, n_samples):
        return 0 if n_samples < self.frame_len else 1 + (n_samples - self.frame_len) // self.shift

    def mel_plan(self):
        """Triangular mel filters as CSR over FFT bins [0, nfft/2): (lo, cnt, ptr, weights)."""
        def mel(f):
            return 1127.0 * math.log(1.0 + f / 700.0)
        nyq = 0.5 * self.sample_frequency
        high = self.high_freq + nyq if self.high_freq <= 0 else self.high_freq
        nb = self.nfft // 2
        bw = self.sample_frequency / self.nfft
        ml, mh = mel(self.low_freq), mel(high)
        delta = (mh - ml) / (self.num_mel_bins + 1)
        mels = [mel(bw * i) for i in range(nb)]
        lo, cnt, ptr, w = [], [], [], []
        for b in range(self.num_mel_bins):
            left, center, right = ml + b * delta, ml + (b + 1) * delta, ml + (b + 2) * delta
            idx = [i for i in range(nb) if left < mels[i] < right]
            lo.append(idx[0] if idx else 0)
            cnt.append(len(idx))
            ptr.append(len(w))
            for i in idx:
                m = mels[i]
                w.append((m - left) / (center - left) if m <= center else (right - m) / (right - center))
        return (np.array(lo, np.int32), np.array(cnt, np.int32), np.array(ptr, np.int32),
                np.array(w, np.float32))


class GpuFrontEnd(object):
    def __init__(self, cfg, device, lctx=1, rctx=1, stride=1):
        if device.type != "cuda":
            raise RuntimeError("pika_amd GpuFrontEnd needs a HIP device (no CPU path)")
        self.cfg, self.device = cfg, device
        self.lctx, self.rctx, self.stride = lctx, rctx, stride
        self.plan = [torch.from_numpy(a).to(device) for a in cfg.mel_plan()]
        self.seed = 0

    def __call__(self, pcms, rates, target_dbs, perturb=True):
        """pcms: list of int16 numpy arrays; rates / target_dbs: per-utterance speed and target
        RMS dB.  Returns (data (B,Tmax,D) f32 on the device, frame lengths list)."""
        cfg, dev = self.cfg, self.device
        lib = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        B = len(pcms)
        n_in = [len(p) for p in pcms]
        n_out = [n if (not perturb or r == 1.0) else int(n / r) for n, r in zip(n_in, rates)]
        in_off = np.concatenate(([0], np.cumsum(n_in))).astype(np.int64)
        out_off = np.concatenate(([0], np.cumsum(n_out))).astype(np.int64)
        frames = [cfg.num_frames(n) for n in n_out]
        fr_off = np.concatenate(([0], np.cumsum(frames))).astype(np.int64)
        host = torch.empty(int(in_off[-1]), dtype=torch.int16).pin_memory()
        host.numpy()[:] = np.concatenate(pcms) if B else np.zeros(0, np.int16)
        with torch.cuda.device(dev):
            pcm_d = host.to(dev, non_blocking=True)
            offs = torch.from_numpy(np.stack([in_off, out_off, fr_off])).to(dev)
            wave = torch.empty(max(int(out_off[-1]), 1), dtype=torch.float32, device=dev)
            if perturb:
                db = torch.tensor(list(target_dbs), dtype=torch.float64, device=dev)
                sumsq = torch.empty(B, dtype=torch.float64, device=dev)
                _lib.check(lib.pika_audio_perturb(pcm_d.data_ptr(), offs[0].data_ptr(), offs[1].data_ptr(),
                                                  db.data_ptr(), B, max(n_out), wave.data_ptr(),
                                                  sumsq.data_ptr(), st), "pika_audio_perturb")
            else:
                wave[:int(out_off[-1])] = pcm_d.float()
            total = int(fr_off[-1])
            feats = torch.empty((max(total, 1), cfg.num_mel_bins), dtype=torch.float32, device=dev)
            self.seed += 1
            _lib.check(lib.pika_fbank(wave.data_ptr(), offs[1].data_ptr(), offs[2].data_ptr(), B, total,
                                      cfg.frame_len, cfg.shift, cfg.nfft, cfg.preemphasis_coefficient,
                                      cfg.dither, self.seed, cfg.num_mel_bins,
                                      *[p.data_ptr() for p in self.plan], feats.data_ptr(), st), "pika_fbank")
            lens = [(f + self.stride - 1) // self.stride for f in frames]
            t_max = max(lens) if lens else 0
            D = cfg.num_mel_bins * (self.lctx + 1 + self.rctx)
            out = torch.zeros((B, max(t_max, 1), D), dtype=torch.float32, device=dev)
            if t_max > 0:
                _lib.check(lib.pika_splice_pad(feats.data_ptr(), offs[2].data_ptr(), B, cfg.num_mel_bins,
                                               self.lctx, self.rctx, self.stride, t_max, out.data_ptr(),
                                               st), "pika_splice_pad")
        self.last_feats, self.last_wave, self.last_offsets = feats, wave, (in_off, out_off, fr_off)
        return out, lens
