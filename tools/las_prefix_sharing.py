"""How much of the LAS token loop is shared between the n-best entries of an utterance?  Decodes the configs[4] bench batch
and counts, per token step, the hypotheses still active and the DISTINCT prefixes among them (forward lists and reversed
lists): the rows a prefix-sharing token loop would have to compute.   python tools/las_prefix_sharing.py"""
import os
import sys
from types import SimpleNamespace

import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
import bench  # noqa: E402

a = SimpleNamespace(batch=64, beam=16, frames=1000, labels=50, vocab=5000, fst=False, las=False, pred_net="transformer",
                    blank_bias=None, fst_scale=0.3)
step, _ = bench.decode_workload(a, torch.device("cuda:0"), 0)
ret, _ = step()
cap = 2 * a.labels
hyps = [[[int(e) for e in h if int(e) != 0][:cap] for h in ret["predictions"][i]] for i in range(a.batch)]
for name, lists in (("forward", hyps), ("reversed", [[h[::-1] for h in row] for row in hyps])):
    act = dist = 0
    L = max(len(h) for row in lists for h in row) + 1
    # rows ordered by the step they diverge from an earlier entry; a prefix of that order is computed per step
    prefix_order = 0
    for t in range(L):
        n_act_t = n_dist_t = 0
        for row in lists:
            alive = [tuple(h[:t]) for h in row if len(h) + 1 > t]
            n_act_t += len(alive)
            n_dist_t += len(set(alive))
        act += n_act_t
        dist += n_dist_t
    # prefix-of-activation-order variant: row computed from its activation step to the END of the loop for its utterance
    waste = 0
    for row in lists:
        Lr = max(len(h) for h in row) + 1
        seen_t = []
        for j, h in enumerate(row):
            # activation step: first t at which no earlier entry shares h[:t]
            tj = 0
            for t in range(len(h) + 2):
                if not any(tuple(g[:t]) == tuple(h[:t]) and len(g) + 1 > t for g in row[:j]):
                    tj = t
                    break
            else:
                tj = len(h) + 1
            seen_t.append(tj)
        waste += sum(max(0, Lr - tj) for tj in seen_t)
    print("%s lists: (step, hypothesis) pairs %d, distinct-prefix pairs %d (%.0f %%); activation-ordered prefix ranges %d (%.0f %%)"
          % (name, act, dist, 100.0 * dist / act, waste, 100.0 * waste / act), flush=True)
