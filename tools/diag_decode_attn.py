"""Diagnostic: the full-width beam golden with the decoder's encoder attention on the exact torch chain (pika_amd.model.hipops.INFER_ATTN = False)
and on the fused fp16 two-term kernel: encoder output error, score differences, and the neighbourhood of every n-best
entry that is not at its reference rank."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pika_amd  # noqa: F401,E402
import test_decode_full as TD  # noqa: E402

z = np.load(TD.GOLD)
for flag in ("0", "1"):
    from pika_amd.model import hipops
    hipops.INFER_ATTN = flag != "0"
    got, enc, d = TD.decode("cuda:0", "fp32")
    es = enc[:, ::7, ::37].float().cpu().numpy()
    rel = np.abs(es - z["enc_sample"]).max() / np.abs(z["enc_sample"]).max()
    B, nb = z["lens"].shape
    print("INFER_ATTN=%s: encoder output rel err %.2e, max |score diff| %.2e" % (
        flag, rel, float(np.abs(got["scores"] - z["scores"]).max())))
    for b in range(B):
        for j in range(nb):
            if not TD.same_entry(got, z, b, j):
                sc = z["scores"][b]
                where = [k for k in range(nb) if TD.same_entry(got, z, b, j, k)]
                print("   utt %d rank %d: reference score %.5f (neighbours %+.5f / %+.5f); ours has it at rank %s; our scores "
                      "around: %s" % (b, j, sc[j], sc[j - 1] - sc[j] if j else float("nan"),
                                      sc[j] - sc[j + 1] if j + 1 < nb else float("nan"), where,
                                      np.round(got["scores"][b, max(j - 1, 0):j + 2], 5).tolist()))
