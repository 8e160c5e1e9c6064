"""Offline Kaldi-feature loader used by decoding (loader/utt_loader.py:16-69,155-237): same
`register` / `dataloader(align_rspec, feats_rspec, dummy, args)` interface; features are spliced,
strided and padded on the GPU by the same kernel as the on-the-fly loader when `args.cuda`."""
import numpy as np
import torch

from . import kaldi_io
from .otf_utt_loader import get_inputdim, splice  # noqa: F401


def register(parser):
    a = parser.add_argument
    a('--lctx', type=int, default=10)
    a('--rctx', type=int, default=10)
    a('--max_len', type=int, default=6000)
    a('--buffer_size', type=int, default=128 * 1024)
    a('--ctc_target', action='store_true')
    a('--batch_first', action='store_true')
    a('--stride', type=int, default=1)
    a('--batch_size', type=int, default=1024)
    a('--queue_size', type=int, default=8)
    a('--padding_tgt', type=int, default=-1)
    a('--feats_dim', type=int, default=40)
    a('--verbose', action='store_true')


def _batch(items, args):
    feats = [splice(f, args.lctx, args.rctx)[::args.stride] for _, _, f in items]
    lens = np.array([f.shape[0] for f in feats], np.int32)
    alis = [a for _, a, _ in items]
    ali_lens = np.array([len(a) for a in alis], np.int32)
    tmax, umax = int(lens.max()), int(ali_lens.max())
    data = np.zeros((len(items), tmax, feats[0].shape[1]), np.float32)
    target = np.full((len(items), umax), args.padding_tgt, np.int64)
    for i, (f, a) in enumerate(zip(feats, alis)):
        data[i, :len(f)] = f
        data[i, len(f):] = f[-1]                                             # :199-201
        target[i, :len(a)] = a
    if not args.batch_first:
        data, target = data.transpose(1, 0, 2).copy(), target.T.copy()
    data, target = torch.from_numpy(data), torch.from_numpy(target).long()
    if getattr(args, "cuda", False):
        data, target = data.cuda(args.local_rank), target.cuda(args.local_rank)
    return data, target, lens, ali_lens


def dataloader(align_rspec, feats_rspec, dummy_args, args):
    """Yields full batches only, then None -- exactly like the reference (a trailing partial
    batch is dropped, loader/utt_loader.py:192,237)."""
    if getattr(args, "ctc_target", False):
        raise NotImplementedError("ctc_utt_generator is not used by the RNN-T path (SURVEY 2.1)")
    items = []
    for (uttid, ali), (uttid2, feats) in zip(kaldi_io.read_int_vectors(align_rspec),
                                             kaldi_io.read_matrices(feats_rspec)):
        assert uttid2 == uttid
        items.append((uttid, ali, feats))
        if len(items) == args.batch_size:
            yield _batch(items, args)
            items = []
    yield None
