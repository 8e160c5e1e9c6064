"""Post-decode steps: n-best reranking against golden output of the reference script
(tests/golden/make_rerank_golden.py ran egs/local/nbest_rerank.py itself), character splitting / key
attachment of eval_transducer.sh:105-124, and the WER counts (Kaldi binary absent: hand-built cases)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_nbest_rerank_matches_reference_script(tmp_path):
    from pika_amd.eval import nbest_rerank as R
    z = np.load(os.path.join(HERE, "golden", "nbest_rerank.npz"))
    for name in ("plain", "las", "las_scaled"):
        fi, fo = tmp_path / (name + ".in"), tmp_path / (name + ".out")
        fi.write_text(str(z[name + "/in"]) + "\n", encoding="utf-8")
        R.main(str(z[name + "/args"]).split() + [str(fi), str(fo)])
        assert fo.read_text(encoding="utf-8") == str(z[name + "/out"]), name


def test_char_split_and_keys():
    from pika_amd.eval.scoring import attach_keys, char_split_text
    assert char_split_text("utt1 你好 [noise] <unk> ab !SIL") == "utt1 你 好 [noise] <unk> a b !SIL"
    assert char_split_text("") == ""
    assert attach_keys(["utt1 3 4 5", "utt2 7"], ["a b <unk> c", ""]) == ["utt1 a b  c", "utt2"]


def test_compute_wer_counts():
    from pika_amd.eval.scoring import align_counts, compute_wer, format_wer
    assert align_counts(list("abcd"), list("abcd")) == (0, 0, 0)
    assert align_counts(list("abcd"), list("abd")) == (0, 1, 0)
    assert align_counts(list("abd"), list("abcd")) == (1, 0, 0)
    assert align_counts(list("abcd"), list("axcd")) == (0, 0, 1)
    assert sum(align_counts(list("kitten"), list("sitting"))) == 3
    assert align_counts([], list("ab")) == (2, 0, 0) and align_counts(list("ab"), []) == (0, 2, 0)
    ref = ["u1 a b c d", "u2 x y", "u3 q"]
    hyp = ["u1 a b d", "u2 x y"]
    t = compute_wer(ref, hyp)                       # mode=present: u3 skipped
    assert (t["words"], t["ins"], t["del"], t["sub"], t["sents"], t["sent_errs"]) == (6, 0, 1, 0, 2, 1)
    assert abs(t["wer"] - 100.0 / 6) < 1e-9 and abs(t["ser"] - 50.0) < 1e-9
    assert format_wer(t).startswith("%WER 16.67 [ 1 / 6, 0 ins, 1 del, 0 sub ]")
    t = compute_wer(ref, hyp, mode="all")
    assert (t["words"], t["del"], t["sents"]) == (7, 2, 3)
