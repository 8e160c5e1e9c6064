"""Decode scenarios shared by the golden generator (reference side) and tests/test_decode.py."""
import numpy as np
import torch

import model_common as C

SCENARIOS = {
    "greedy": dict(beam=1, n_best=1, sm_scale=1.0, max_len=None),
    "beam4": dict(beam=4, n_best=4, sm_scale=0.8, max_len=None),
    "beam4_nbest2": dict(beam=4, n_best=2, sm_scale=1.0, max_len=None),
    "beam3_short": dict(beam=3, n_best=3, sm_scale=1.0, max_len=7),   # exercises len > max_len
    "beam8": dict(beam=8, n_best=8, sm_scale=0.8, max_len=None),
}


SHARP, BLANK_BIAS = 8.0, 8.0


def tweak(net):
    """Random weights give near-uniform posteriors (searches then emit only labels, or only
    blanks).  Sharpen fc2 and nudge the blank bias so that hypotheses mix blank and label steps
    and finish by running out of frames -- identically on the reference and on our side."""
    with torch.no_grad():
        net.fc2.weight *= SHARP
        net.fc2.bias[0] += BLANK_BIAS
    return net


def inputs():
    x, _, _, _ = C.inputs()
    # encoder frames after the 6-layer TDNN of the tiny model: 17; ragged valid lengths
    x_len = torch.tensor([17, 15, 12, 17])
    return x, x_len


def max_len(cfg, x_len):
    if cfg["max_len"] is None:
        return [int(v) + 100 for v in x_len]       # decode_transducer.py:132-133
    return [cfg["max_len"]] * len(x_len)


def pack(preds, scores):
    """Ragged n-best lists -> arrays (-9 padding)."""
    B, nb = len(preds), len(preds[0])
    L = max(len(h) for p in preds for h in p) if B else 0
    arr = np.full((B, nb, max(L, 1)), -9, np.int64)
    lens = np.zeros((B, nb), np.int64)
    sc = np.zeros((B, nb), np.float64)
    for b in range(B):
        for j in range(nb):
            h = [int(e) for e in preds[b][j]]
            arr[b, j, :len(h)] = h
            lens[b, j] = len(h)
            sc[b, j] = float(scores[b][j])
    return {"hyps": arr, "lens": lens, "scores": sc}
