"""TEST-ONLY preload for `pika_amd.launch` (BASELINE.json configs[0]: CPU plumbing, world_size 1):
lets the UNCHANGED reference training script run in a container without a GPU by (a) making the
CUDA calls it hard-codes no-ops and (b) standing the CPU ORACLES in for the two GPU-only product
ops (RNN-T loss, feature front end).  Test infrastructure: never imported by the product."""
import numpy as np
import torch

from oracle import rnnt as O
from oracle import fbank_ref as F

torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda *a, **k: None
torch.cuda.manual_seed = lambda *a, **k: None
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self


class _OracleLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lp, labels, tl, ul):
        costs, grads = O.rnnt_loss(lp.detach().numpy(), labels.numpy(), tl.numpy(), ul.numpy(), dtype=np.float32)
        ctx.save_for_backward(torch.from_numpy(grads))
        return torch.from_numpy(costs)

    @staticmethod
    def backward(ctx, g):
        (grads,) = ctx.saved_tensors
        return grads * g.view(-1, 1, 1, 1), None, None, None


import pika_amd.rnnt as R  # noqa: E402


class RNNTLoss(object):
    def __init__(self, blank=0, reduction="sum"):
        pass

    def apply(self, lp, labels, tl, ul):
        return _OracleLoss.apply(lp, labels, tl.int(), ul.int())


R.RNNTLoss = RNNTLoss

import pika_amd.loader.otf_utt_loader as L  # noqa: E402


class _CpuFrontEnd(object):
    stream = None          # no side stream: the loader then assembles batches on the consumer thread

    def __init__(self, cfg, dev, lctx, rctx, stride, base_seed=0, side_stream=False):
        self.cfg, self.lctx, self.rctx, self.stride = cfg, lctx, rctx, stride

    def __call__(self, pcms, rates, dbs):
        feats = [F.splice(F.kaldi_fbank(F.perturb(p, r, d).astype(np.float64), num_bins=self.cfg.num_mel_bins,
                                        low=self.cfg.low_freq, high=self.cfg.high_freq).astype(np.float32),
                          self.lctx, self.rctx)[::self.stride] for p, r, d in zip(pcms, rates, dbs)]
        lens = [f.shape[0] for f in feats]
        out = np.zeros((len(feats), max(lens), feats[0].shape[1]), np.float32)
        for i, f in enumerate(feats):
            out[i, :len(f)] = f
            out[i, len(f):] = f[-1]
        return torch.from_numpy(out), lens


L.GpuFrontEnd = _CpuFrontEnd
_real_device = torch.device
L.torch = type("T", (), {"device": lambda *a, **k: _real_device("cpu"), "IntTensor": torch.IntTensor,
                         "from_numpy": torch.from_numpy, "tensor": torch.tensor, "int32": torch.int32})

# the MBR script builds tensors through the legacy CUDA type constructors (train_transducer_mbr_bmuf_otfaug.py:167-199)
torch.cuda.FloatTensor = torch.FloatTensor
torch.cuda.LongTensor = torch.LongTensor
