"""Scoring after reranking (reference egs/eval_transducer.sh:105-127), without perl or Kaldi binaries.

* `char_split_text`: the perl one-liner of :109-124 -- every word of a Kaldi `text` line is split into
  characters unless it looks like a tag (`[...]`, `<...>`) or contains `!SIL`.
* `attach_keys`: `paste label.ark raw.hyp | awk '{$2=" "; print}' | sed 's/<unk>//g'` (:108): the utterance key
  of the label archive line followed by the hypothesis characters.
* `compute_wer`: Kaldi `compute-wer --text --mode=present ark:ref ark:hyp` -- word-level Levenshtein alignment
  of every hypothesis against the reference with the same key (hypothesis keys absent from the reference are an
  error; references without a hypothesis are skipped in mode `present`), totals
  %WER = 100 (ins + del + sub) / ref words, %SER = sentences with any error.
  PARITY UNPINNED: the Kaldi binary is not in this image; the output format follows Kaldi's documented one and
  the counts are pinned by hand-built cases (tests/test_eval.py).
"""
import re

_TAG1, _TAG2 = re.compile(r"\[.*\]"), re.compile(r"\<.*\>")


def char_split_text(line):
    f = line.split()
    if not f:
        return ""
    out = [f[0]]
    for s in f[1:]:
        if _TAG1.search(s) or _TAG2.search(s) or "!SIL" in s:
            out.append(s)
        else:
            out.extend(list(s))
    return " ".join(out)


def attach_keys(label_lines, hyp_lines):
    """One scoring line per utterance: `key  c1 c2 ...` (the label ids of the archive line are dropped)."""
    out = []
    for lab, hyp in zip(label_lines, hyp_lines):
        key = lab.split()[0]
        out.append((key + " " + hyp.strip()).replace("<unk>", "").rstrip())
    return out


def align_counts(ref, hyp):
    """(ins, del, sub) of a minimum-edit-distance alignment (ties: substitutions preferred, as Kaldi's
    LevenshteinEditDistance does: it minimises the total and reports one optimal alignment's split)."""
    n, m = len(ref), len(hyp)
    # cost, ins, del, sub
    prev = [(j, j, 0, 0) for j in range(m + 1)]
    for i in range(1, n + 1):
        cur = [(i, 0, i, 0)]
        for j in range(1, m + 1):
            if ref[i - 1] == hyp[j - 1]:
                best = prev[j - 1]
            else:
                c, a, b, s = prev[j - 1]
                best = (c + 1, a, b, s + 1)
            c, a, b, s = prev[j]
            if c + 1 < best[0]:
                best = (c + 1, a, b + 1, s)
            c, a, b, s = cur[j - 1]
            if c + 1 < best[0]:
                best = (c + 1, a + 1, b, s)
            cur.append(best)
        prev = cur
    _, ins, dele, sub = prev[m]
    return ins, dele, sub


def compute_wer(ref_lines, hyp_lines, mode="present"):
    refs = {}
    for line in ref_lines:
        f = line.split()
        if f:
            refs[f[0]] = f[1:]
    tot = {"words": 0, "ins": 0, "del": 0, "sub": 0, "sents": 0, "sent_errs": 0}
    seen = set()
    for line in hyp_lines:
        f = line.split()
        if not f:
            continue
        key, hyp = f[0], f[1:]
        if key not in refs:
            raise KeyError("compute_wer: hypothesis key %r has no reference" % key)
        seen.add(key)
        ins, dele, sub = align_counts(refs[key], hyp)
        tot["words"] += len(refs[key]); tot["ins"] += ins; tot["del"] += dele; tot["sub"] += sub
        tot["sents"] += 1
        tot["sent_errs"] += int(ins + dele + sub > 0)
    if mode == "all":      # references without a hypothesis count as all-deleted
        for key, ref in refs.items():
            if key not in seen:
                tot["words"] += len(ref); tot["del"] += len(ref); tot["sents"] += 1; tot["sent_errs"] += int(len(ref) > 0)
    errs = tot["ins"] + tot["del"] + tot["sub"]
    tot["wer"] = 100.0 * errs / tot["words"] if tot["words"] else 0.0
    tot["ser"] = 100.0 * tot["sent_errs"] / tot["sents"] if tot["sents"] else 0.0
    return tot


def format_wer(t):
    errs = t["ins"] + t["del"] + t["sub"]
    return ("%%WER %.2f [ %d / %d, %d ins, %d del, %d sub ]\n%%SER %.2f [ %d / %d ]\nScored %d sentences, 0 not present in hyp."
            % (t["wer"], errs, t["words"], t["ins"], t["del"], t["sub"], t["ser"], t["sent_errs"], t["sents"], t["sents"]))


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="char-split a Kaldi text file, attach keys to reranked hypotheses, score")
    ap.add_argument("text"); ap.add_argument("label_ark"); ap.add_argument("raw_hyp")
    ap.add_argument("--mode", default="present", choices=["present", "all"])
    a = ap.parse_args(argv)
    with open(a.text, encoding="utf-8") as f:
        ref = [char_split_text(l) for l in f]
    with open(a.label_ark, encoding="utf-8") as f1, open(a.raw_hyp, encoding="utf-8") as f2:
        hyp = attach_keys(list(f1), list(f2))
    print(format_wer(compute_wer(ref, hyp, a.mode)))


if __name__ == "__main__":
    main()
