"""On-the-fly utterance loader with the reference's interface
(loader/otf_utt_loader.py:61-299: `register`, `get_inputdim`, `dataloader`).

Host side (a producer thread per worker, bounded queue -- same threading model): read int16 PCM
at the `.mrk` offsets, draw the speed rate with `random.randint` and the target level with
`np.random.uniform` in the reference's order (:221-223), collect `batch_size` utterances.
Device side (consumer): ONE pinned-buffer upload per batch, then perturbation + fbank + splice +
padding kernels (frontend.py).  Batches come back with `data` already on the GPU (the training
script's `.cuda(local_rank)` is then a no-op) and targets/lengths on the CPU like the reference.
The T*U filter (:247) needs only the frame COUNT, which is known on the host from the sample count.
"""
import queue
from random import randint
from threading import Thread

import numpy as np
import torch

from . import kaldi_io
from .frontend import FbankConfig, GpuFrontEnd


def register(parser):
    """Same flags, defaults and help-less semantics as loader/otf_utt_loader.py:68-114."""
    a = parser.add_argument
    a('--lctx', type=int, default=10)
    a('--rctx', type=int, default=10)
    a('--max_len', type=int, default=6000)
    a('--num_workers', type=int, default=5)
    a('--sample_rate', type=int, default=16000)
    a('--buffer_size', type=int, default=128 * 1024)
    a('--batch_first', action='store_true')
    a('--reverse_labels', action='store_true')
    a('--feat_config', type=str, default=None)
    a('--stride', type=int, default=1)
    a('--batch_size', type=int, default=1024)
    a('--SOS', type=int, default=-1)
    a('--EOS', type=int, default=-1)
    a('--queue_size', type=int, default=8)
    a('--TU_limit', type=int, default=15000)
    a('--padding_tgt', type=int, default=-1)
    a('--feats_dim', type=int, default=40)
    a('--snr_range', type=str, default='')
    a('--gain_range', type=str, default='55,10')
    a('--speed_rate', type=str, default='0.9,1.0,1.1')
    a('--verbose', action='store_true')


def get_inputdim(args):
    return args.feats_dim * (args.lctx + 1 + args.rctx)


def splice(feats, lctx, rctx):
    """numpy splice kept for callers that import it (loader/utt_loader.py:12)."""
    n = feats.shape[0]
    pad = np.concatenate([np.repeat(feats[:1], lctx, 0), feats, np.repeat(feats[-1:], rctx, 0)])
    return np.concatenate([pad[i:i + n] for i in range(lctx + 1 + rctx)], axis=1).astype(np.float32)


def put_thread(q, generator, *gen_args):
    for item in generator(*gen_args):
        q.put(item)
        if item is None:
            break


def host_batches(data_triplets, cfg, args):
    """Host half of otf_utt_generator (:165-299): yields raw batches
    (pcms, rates, target_dbs, labels, frame_lens) or None at the end."""
    speed_rate = [float(r) for r in args.speed_rate.split(',')]
    gain_lo, gain_hi = [-float(g) for g in args.gain_range.split(',')]
    batch, batch_idx = [], 0
    for mrk_fn, seq_fn, ali_rspec in data_triplets:
        with open(seq_fn, 'rb') as seq:
            for (uttid, off, nbytes), (uttid1, ali) in zip(kaldi_io.read_mrk(mrk_fn),
                                                           kaldi_io.read_int_vectors(ali_rspec)):
                assert uttid == uttid1                                       # :212
                pcm = kaldi_io.read_pcm(seq, off, nbytes)
                spr = speed_rate[randint(0, len(speed_rate) - 1)]            # :221
                target_db = np.random.uniform(gain_lo, gain_hi)              # :223
                n_out = len(pcm) if spr == 1.0 else int(len(pcm) / spr)
                frames = cfg.num_frames(n_out)
                if args.reverse_labels:
                    ali = ali[::-1]
                if args.SOS >= 0:
                    ali = np.concatenate(([args.SOS], ali))
                if args.EOS >= 0:
                    ali = np.concatenate((ali, [args.EOS]))
                utt_len = frames // args.stride + int(frames % args.stride != 0)
                if ali.shape[0] * utt_len // 3 <= args.TU_limit and frames > 0:   # :247
                    batch.append((pcm, spr, target_db, np.asarray(ali, np.int32), utt_len))
                batch_idx += 1
                if batch_idx == args.batch_size:
                    yield batch
                    batch, batch_idx = [], 0
    yield None


def assemble(batch, frontend, args):
    """Device half: features for the kept utterances + padded targets (:253-289)."""
    if not batch:
        return None, None, torch.IntTensor([0]), torch.IntTensor([0])
    pcms, rates, dbs, alis, lens = zip(*batch)
    data, flens = frontend(list(pcms), list(rates), list(dbs))
    assert list(flens) == list(lens)
    umax = max(len(a) for a in alis)
    target = np.full((len(batch), umax), args.padding_tgt, np.int32)
    for i, a in enumerate(alis):
        target[i, :len(a)] = a
    st = getattr(frontend, "stream", None)
    if not args.batch_first:
        if st is not None:
            # the feature kernels were issued on the front end's own stream: the transposing copy must run behind them
            # on THAT stream, and `ready` must cover it (the consumer only waits on the event)
            with torch.cuda.stream(st):
                data = data.transpose(0, 1).contiguous()
                frontend.ready = torch.cuda.Event()
                frontend.ready.record(st)
        else:
            data = data.transpose(0, 1).contiguous()
        target = target.T.copy()
    ali_lens = np.array([len(a) for a in alis], np.int32)
    if st is not None and getattr(args, "device_targets", True):
        # targets and lengths travel with the batch (one pinned upload on the front end's stream): the script's own
        # `.cuda(local_rank)` on them (train_transducer_bmuf_otfaug.py:79-85) is then a no-op instead of three
        # synchronous copies per step
        t_d, l_d, a_d = frontend.upload_int32([target, np.asarray(lens, np.int32), ali_lens])
        return data, t_d, l_d, a_d
    return (data, torch.from_numpy(target), torch.tensor(lens, dtype=torch.int32), torch.from_numpy(ali_lens))


class DevicePrefetcher(object):
    """Runs the device half of the loader (`assemble`: staging, upload, perturbation, fbank, splice) on the front
    end's side stream from its own thread, `depth` batches ahead of the consumer.  The consumer's stream waits on
    the batch's HIP event when it takes the batch (no host wait): the front end of batch n+1 overlaps step n.
    Host batches come from `source` (a queue of raw batches; `None` x n_end marks the end)."""

    _END = object()

    def __init__(self, source, n_end, frontend, args, depth=2):
        self.frontend, self.args = frontend, args
        self.out = queue.Queue(max(int(depth), 1))
        self.thread = Thread(target=self._run, args=(source, n_end))
        self.thread.daemon = True
        self.thread.start()

    def _run(self, source, n_end):
        done = 0
        try:
            with torch.cuda.device(self.frontend.device):
                while done < n_end:
                    item = source.get()
                    if item is None:
                        done += 1
                        continue
                    batch = assemble(item, self.frontend, self.args)
                    self.out.put((batch, self.frontend.ready))
        except BaseException as e:     # surfaces in the consumer, not in a dead thread
            self.out.put((e, None))
            return
        self.out.put((self._END, None))

    def __iter__(self):
        while True:
            batch, ready = self.out.get()
            if batch is self._END:
                break
            if isinstance(batch, BaseException):
                raise batch
            if ready is not None:
                torch.cuda.current_stream(self.frontend.device).wait_event(ready)
                for t in batch:
                    if t is not None and t.is_cuda:
                        t.record_stream(torch.cuda.current_stream(self.frontend.device))
            yield batch
        self.thread.join()


def dataloader(data_lst, rir, noise, args, frontend=None):
    """Generator of (data, target, lens, ali_lens) batches (:116-163)."""
    cfg = FbankConfig.from_file(args.feat_config)
    triplets = []
    with open(data_lst, 'r', encoding='utf-8') as f:
        for line in f:
            p = line.split()
            triplets.append((p[0], p[1], p[2]))
    per = (len(triplets) + args.num_workers - 1) // args.num_workers
    parts = [triplets[i:i + per] for i in range(0, len(triplets), per)]
    assert len(parts) == args.num_workers                                    # :135
    if frontend is None:
        rank = getattr(args, "local_rank", 0) or 0
        dev = torch.device("cuda", rank)
        frontend = GpuFrontEnd(cfg, dev, args.lctx, args.rctx, args.stride,
                               base_seed=(int(getattr(args, "seed", 0) or 0) * 64 + rank), side_stream=True)
    q = queue.Queue(args.queue_size)
    threads = [Thread(target=put_thread, args=(q, host_batches, part, cfg, args)) for part in parts]
    for t in threads:
        t.daemon = True
        t.start()
    if getattr(frontend, "stream", None) is not None:
        # device half on the front end's own stream, two batches ahead of the training step
        for batch in DevicePrefetcher(q, args.num_workers, frontend, args):
            yield batch
    else:
        done = 0
        while True:
            item = q.get()
            if item is None:
                done += 1
                if done == args.num_workers:
                    break
                continue
            yield assemble(item, frontend, args)
    for t in threads:
        t.join()
