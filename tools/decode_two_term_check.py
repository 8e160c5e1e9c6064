"""How close is the two-term decode (decode_precision "bf16x3") to the reference n-best lists of the full-width golden?
Runs the strict criterion of tests/test_decode_full.py (every entry separated from its neighbours by > 1e-3 in score at
its reference rank) for both modes and prints the statistics.   python tools/decode_two_term_check.py"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_decode_full as T  # noqa: E402

z = np.load(T.GOLD)
for prec in ("fp32", "fp32-exact", "bf16x3"):
    got, enc, d = T.decode("cuda:0", prec)
    es = enc[:, ::7, ::37].float().cpu().numpy()
    rel = np.abs(es - z["enc_sample"]).max() / np.abs(z["enc_sample"]).max()
    B, nb = z["lens"].shape
    n_same = n_sep = n_sep_diff = 0
    worst = 0.0
    for b in range(B):
        sc = z["scores"][b]
        for j in range(nb):
            sep = (j == 0 or sc[j - 1] - sc[j] > 1e-3) and (j == nb - 1 or sc[j] - sc[j + 1] > 1e-3)
            same = T.same_entry(got, z, b, j)
            n_same += int(same); n_sep += int(sep); n_sep_diff += int(sep and not same)
            if same:
                worst = max(worst, abs(float(got["scores"][b, j] - sc[j])))
    top1 = sum(int(T.same_entry(got, z, b, 0)) for b in range(B))
    print("%s (terms %d): encoder rel err %.2e; top-1 identical %d/%d; entries at reference rank %d/%d; separated entries "
          "%d of which differing %d; max |score diff| on identical entries %.2e; search %.3f s"
          % (prec, d.decode_terms, rel, top1, B, n_same, B * nb, n_sep, n_sep_diff, worst, d.timing["search_s"]), flush=True)
