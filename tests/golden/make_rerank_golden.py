"""Golden for pika_amd/eval/nbest_rerank.py: runs the REFERENCE script (egs/local/nbest_rerank.py, unchanged,
as a subprocess) on seeded synthetic n-best files and stores inputs + outputs in tests/golden/nbest_rerank.npz.
Run in the build container (needs /root/reference):  python tests/golden/make_rerank_golden.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference/egs/local/nbest_rerank.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def synth(rng, n_utt, nbest, las):
    chars = list("abcdefgh") + ["<unk>", "你", "好"]
    lines = []
    for _ in range(n_utt):
        for _ in range(nbest):
            L = int(rng.integers(0, 9))
            hyp = "".join(rng.choice(chars) for _ in range(L))
            f = [hyp] if L else []
            f.append("%.4f" % (-rng.random() * 30))
            if las:
                k = max(L + 1, 1)
                f += ["%.4f" % (-rng.random() * 5) for _ in range(2 * k if L else 2)]
            lines.append(" ".join(f))
    # ties and an all-empty group
    lines[0:nbest] = ["ab -3.0" + (" -1 -1 -1 -1 -1 -1" if las else "")] * nbest
    return lines


def main():
    rng = np.random.default_rng(11)
    out = {}
    for name, las, nbest, extra in (("plain", False, 4, []), ("las", True, 3, []),
                                    ("las_scaled", True, 5, ["--rnnt_score_scale", "0.7", "--las_fw_score_scale", "0.2",
                                                             "--las_bw_score_scale", "0.5"])):
        lines = synth(rng, 12, nbest, las)
        with tempfile.TemporaryDirectory() as d:
            fi, fo = os.path.join(d, "in"), os.path.join(d, "out")
            open(fi, "w", encoding="utf-8").write("\n".join(lines) + "\n")
            cmd = [sys.executable, REF, "--nbest", str(nbest)] + (["--las_rescore"] if las else []) + extra + [fi, fo]
            subprocess.run(cmd, check=True)
            got = open(fo, encoding="utf-8").read()
        out[name + "/in"] = np.array("\n".join(lines))
        out[name + "/out"] = np.array(got)
        out[name + "/args"] = np.array(" ".join(["--nbest", str(nbest)] + (["--las_rescore"] if las else []) + extra))
    np.savez(os.path.join(HERE, "nbest_rerank.npz"), **out)
    print("wrote", os.path.join(HERE, "nbest_rerank.npz"))


if __name__ == "__main__":
    main()
