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
    """One call = one batch: pinned host staging -> ONE H2D copy -> perturbation, fbank, splice/pad kernels.

    Host staging is a persistent ring of pinned buffers (a buffer is re-used only after the HIP event recorded
    behind its upload has completed; nothing is allocated or pinned per batch once the ring has grown to the batch
    size).  Sample data, the three offset tables and the target levels travel in the SAME buffer, so a batch costs
    one asynchronous copy.  With `side_stream=True` all device work is issued on the front end's own HIP stream and
    `ready` (a HIP event) marks its end: the consumer's stream waits on the event (`wait_ready`), the host never
    does -- batch n+1 is produced while step n computes (otf_utt_loader.DevicePrefetcher drives it from a thread).

    Dither noise is a counter-based generator keyed by (base_seed, front-end instance, batch number): a new front
    end per epoch (what `dataloader` does) or per rank never replays an earlier sequence."""

    _instances = 0

    def __init__(self, cfg, device, lctx=1, rctx=1, stride=1, base_seed=0, ring=3, side_stream=False):
        if device.type != "cuda":
            raise RuntimeError("pika_amd GpuFrontEnd needs a HIP device (no CPU path)")
        self.cfg, self.device = cfg, device
        self.lctx, self.rctx, self.stride = lctx, rctx, stride
        self.plan = [torch.from_numpy(a).to(device) for a in cfg.mel_plan()]
        GpuFrontEnd._instances += 1
        self.seed_hi = ((int(base_seed) & 0x7FFFFF) << 40) | ((GpuFrontEnd._instances & 0xFF) << 32)
        self.batch_no = 0
        self._ring = [None] * max(int(ring), 1)     # [pinned uint8 tensor, event of its last upload]
        self._slot = 0
        self._ring_small = [None] * (2 * max(int(ring), 1))   # the same for the batch's label / length arrays
        self._slot_small = 0
        self.stream = torch.cuda.Stream(device) if side_stream else None
        self.ready = None                           # event behind the last batch's kernels
        self.host_seconds, self.batches = 0.0, 0    # host time spent inside __call__ (staging + launches)

    @property
    def seed(self):
        return self.seed_hi | (self.batch_no & 0xFFFFFFFF)

    def _stage(self, nbytes):
        """Next pinned buffer of the ring, at least nbytes long, free of in-flight uploads."""
        i = self._slot
        self._slot = (i + 1) % len(self._ring)
        ent = self._ring[i]
        if ent is not None:
            ent[1].synchronize()        # only blocks when the device is a whole ring behind
        if ent is None or ent[0].numel() < nbytes:
            cap = max(int(nbytes * 1.25), 1 << 16)
            # blocking=True: a loader thread that is a whole ring ahead SLEEPS on the event instead of spinning on a core
            # the eight ranks of a node share
            ent = [torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.cuda.Event(blocking=True)]
            self._ring[i] = ent
        return ent

    def upload_int32(self, arrays):
        """The batch's small integer arrays (padded targets, frame and label lengths) as device tensors: packed into ONE
        pinned buffer of a ring, ONE asynchronous copy on the front end's stream, covered by `ready` -- the training
        thread then has no synchronous host-to-device copy left (a pageable `.cuda()` waits, spinning, for everything
        queued before it: 40 ms of CPU per step).  Returns int32 device tensors shaped like the inputs."""
        arrays = [np.ascontiguousarray(a, dtype=np.int32) for a in arrays]
        sizes = [a.size for a in arrays]
        total = max(sum(sizes), 1)
        i = self._slot_small
        self._slot_small = (i + 1) % len(self._ring_small)
        ent = self._ring_small[i]
        if ent is not None:
            ent[1].synchronize()
        if ent is None or ent[0].numel() < total:
            ent = [torch.empty(max(2 * total, 4096), dtype=torch.int32).pin_memory(), torch.cuda.Event(blocking=True)]
            self._ring_small[i] = ent
        host = ent[0].numpy()
        off = 0
        for a in arrays:
            host[off:off + a.size] = a.reshape(-1)
            off += a.size
        dev = self.device
        st_t = self.stream or torch.cuda.current_stream(dev)
        with torch.cuda.device(dev), torch.cuda.stream(st_t):
            d = torch.empty(total, dtype=torch.int32, device=dev)
            d.copy_(ent[0][:total], non_blocking=True)
            ent[1].record(st_t)
            if self.stream is not None:
                self.ready = torch.cuda.Event()
                self.ready.record(st_t)
                d.record_stream(torch.cuda.current_stream(dev))
        out, off = [], 0
        for a, n in zip(arrays, sizes):
            out.append(d[off:off + n].view(a.shape))
            off += n
        return out

    def wait_ready(self, stream=None):
        """Make `stream` (default: the current one) wait for the last batch; no host wait."""
        if self.ready is not None:
            (stream or torch.cuda.current_stream(self.device)).wait_event(self.ready)

    def __call__(self, pcms, rates, target_dbs, perturb=True):
        """pcms: list of int16 numpy arrays; rates / target_dbs: per-utterance speed and target
        RMS dB.  Returns (data (B,Tmax,D) f32 on the device, frame lengths list)."""
        import time
        t_host = time.perf_counter()
        cfg, dev = self.cfg, self.device
        lib = _lib.lib()
        B = len(pcms)
        n_in = [len(p) for p in pcms]
        n_out = [n if (not perturb or r == 1.0) else int(n / r) for n, r in zip(n_in, rates)]
        in_off = np.concatenate(([0], np.cumsum(n_in))).astype(np.int64)
        out_off = np.concatenate(([0], np.cumsum(n_out))).astype(np.int64)
        frames = [cfg.num_frames(n) for n in n_out]
        fr_off = np.concatenate(([0], np.cumsum(frames))).astype(np.int64)
        # staging layout (bytes): [3 x (B+1) i64 offsets][B f64 target dB][int16 samples]
        n_samp = int(in_off[-1])
        o_db = 3 * (B + 1) * 8
        o_pcm = o_db + B * 8
        nbytes = o_pcm + 2 * n_samp
        host, up_ev = self._stage(nbytes)
        hv = host.numpy()
        hv[:o_db].view(np.int64).reshape(3, B + 1)[:] = (in_off, out_off, fr_off)
        hv[o_db:o_pcm].view(np.float64)[:] = np.asarray(list(target_dbs), np.float64)[:B] if B else 0.0
        if B:
            np.concatenate(pcms, out=hv[o_pcm:nbytes].view(np.int16))
        cur = torch.cuda.current_stream(dev)
        st_t = self.stream or cur
        with torch.cuda.device(dev), torch.cuda.stream(st_t):
            st = st_t.cuda_stream
            stage_d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            stage_d.copy_(host[:nbytes], non_blocking=True)
            up_ev.record(st_t)
            base = stage_d.data_ptr()
            p_in, p_out, p_fr = base, base + (B + 1) * 8, base + 2 * (B + 1) * 8
            wave = torch.empty(max(int(out_off[-1]), 1), dtype=torch.float32, device=dev)
            if perturb:
                sumsq = torch.empty(B, dtype=torch.float64, device=dev)
                _lib.check(lib.pika_audio_perturb(base + o_pcm, p_in, p_out, base + o_db, B, max(n_out), wave.data_ptr(),
                                                  sumsq.data_ptr(), st), "pika_audio_perturb")
            else:
                wave[:int(out_off[-1])] = stage_d[o_pcm:nbytes].view(torch.int16).float()
            total = int(fr_off[-1])
            feats = torch.empty((max(total, 1), cfg.num_mel_bins), dtype=torch.float32, device=dev)
            self.batch_no += 1
            _lib.check(lib.pika_fbank(wave.data_ptr(), p_out, p_fr, B, total,
                                      cfg.frame_len, cfg.shift, cfg.nfft, cfg.preemphasis_coefficient,
                                      cfg.dither, self.seed, cfg.num_mel_bins,
                                      *[p.data_ptr() for p in self.plan], feats.data_ptr(), st), "pika_fbank")
            lens = [(f + self.stride - 1) // self.stride for f in frames]
            t_max = max(lens) if lens else 0
            D = cfg.num_mel_bins * (self.lctx + 1 + self.rctx)
            out = torch.zeros((B, max(t_max, 1), D), dtype=torch.float32, device=dev)
            if t_max > 0:
                _lib.check(lib.pika_splice_pad(feats.data_ptr(), p_fr, B, cfg.num_mel_bins,
                                               self.lctx, self.rctx, self.stride, t_max, out.data_ptr(),
                                               st), "pika_splice_pad")
            if self.stream is not None:
                self.ready = torch.cuda.Event()
                self.ready.record(st_t)
                out.record_stream(cur)      # allocated on the side stream, consumed on the caller's
        self.last_feats, self.last_wave, self.last_offsets = feats, wave, (in_off, out_off, fr_off)
        self.host_seconds += time.perf_counter() - t_host
        self.batches += 1
        return out, lens
