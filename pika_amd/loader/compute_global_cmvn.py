"""Global CMVN statistics over a training list (reference utils/compute_global_cmvn.py:16-73, which needs PyKaldi).

    python -m pika_amd.loader.compute_global_cmvn [--cmn] [--sample_rate 16000] --feat_config fbank.conf \\
        [--feat_dim 80] data.lst cmvn.stats

Same protocol as the reference: every utterance of every `.mrk/.seq` pair of `data.lst` is speed-perturbed by a rate
drawn with `random.randint` from (0.9, 1.0, 1.1), normalised to a level drawn with `np.random.uniform(-55, -10)` dB
(same draws in the same order), converted to filter-bank features with the recipe's fbank.conf, optionally
mean-normalised per utterance (--cmn), and accumulated into the 2 x (D+1) Kaldi statistics matrix the training
scripts read (`--cmvn_stats`, train_transducer_bmuf_otfaug.py:340-356), written in Kaldi text format.
Perturbation and filter banks run on the GPU front end of the loader (pika_audio_perturb + pika_fbank), in batches."""
import argparse
from random import randint

import numpy as np
import torch

from . import kaldi_io
from .frontend import FbankConfig, GpuFrontEnd

SPEED_RATES = [0.9, 1.0, 1.1]


def iter_utterances(data_lst):
    """int16 signals in list order (compute_global_cmvn.py:44-55: byte counts are made even)."""
    with open(data_lst, "r", encoding="utf-8") as f:
        for line in f:
            p = line.split()
            if len(p) < 2:
                continue
            with open(p[1], "rb") as seq:
                for _, offset, num_bytes in kaldi_io.read_mrk(p[0]):
                    yield kaldi_io.read_pcm(seq, offset, num_bytes)


def compute(data_lst, feat_config, feat_dim=80, cmn=False, device=None, batch=32, front_end=None):
    cfg = FbankConfig.from_file(feat_config)
    if cfg.num_mel_bins != feat_dim:
        raise ValueError("--feat_dim %d but %s has num-mel-bins %d" % (feat_dim, feat_config, cfg.num_mel_bins))
    fe = front_end or GpuFrontEnd(cfg, device or torch.device("cuda", torch.cuda.current_device()), 0, 0, 1)
    stats = np.zeros((2, feat_dim + 1))

    def flush(pcms, rates, dbs):
        data, lens = fe(pcms, rates, dbs)
        data = data.double().cpu().numpy()
        for i, n in enumerate(lens):
            x = data[i, :n]
            if cmn:
                x = x - x.mean(axis=0)
            stats[0, :-1] += x.sum(0)
            stats[1, :-1] += (x * x).sum(0)
            stats[0, -1] += n

    pcms, rates, dbs = [], [], []
    for pcm in iter_utterances(data_lst):
        pcms.append(pcm)
        rates.append(SPEED_RATES[randint(0, len(SPEED_RATES) - 1)])      # :56
        dbs.append(np.random.uniform(-55, -10))                          # :59
        if len(pcms) == batch:
            flush(pcms, rates, dbs)
            pcms, rates, dbs = [], [], []
    if pcms:
        flush(pcms, rates, dbs)
    return stats


def main(argv=None):
    ap = argparse.ArgumentParser(description="accumulate global mean / variance statistics of filter-bank features over a training list")
    ap.add_argument("data_lst")
    ap.add_argument("cmvn_stats")
    ap.add_argument("--cmn", action="store_true")
    ap.add_argument("--sample_rate", type=int, default=16000)
    ap.add_argument("--feat_config", type=str, default=None)
    ap.add_argument("--feat_dim", type=int, default=80)
    args, _ = ap.parse_known_args(argv)
    kaldi_io.write_text_matrix(args.cmvn_stats, compute(args.data_lst, args.feat_config, args.feat_dim, args.cmn))


if __name__ == "__main__":
    main()
