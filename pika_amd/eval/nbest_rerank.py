"""N-best reranking: same command line, same output file as the reference's egs/local/nbest_rerank.py:8-40.

Input: for every utterance `nbest` consecutive lines, each `hyp rnnt_score [las scores ...]` (hyp = the label
string with <unk> for spaces, missing when the hypothesis is empty).  With --las_rescore the scores after the
RNN-T score are per-token forward LAS scores followed by as many backward ones; a line of <= 3 fields is an empty
hypothesis with exactly one forward and one backward score.  Every hypothesis gets
    score = a * rnnt + [b * sum(fw) + c * sum(bw)],  cost = -score / max(len(hyp), 0.001)
(len in characters after <unk> -> ' '); the lowest cost wins (first on ties: stable sort) and is written as
space-separated characters.

    python -m pika_amd.eval.nbest_rerank [--las_rescore] --nbest N in_hyp out_hyp
"""
import argparse


def score_line(fields, las_rescore, a=1.0, b=0.3, c=0.7):
    """(hyp, score) of one split input line."""
    empty_len = 3 if las_rescore else 1
    if len(fields) <= empty_len:
        hyp = ""
        score = a * float(fields[0])
        if las_rescore:
            score += b * float(fields[1])
            score += c * float(fields[2])
        return hyp, score
    hyp = fields[0].replace("<unk>", " ")
    score = a * float(fields[1])
    if las_rescore:
        n = len(fields) - 2
        score += b * sum(float(s) for s in fields[2:2 + n // 2])
        score += c * sum(float(s) for s in fields[2 + n // 2:])
    return hyp, score


def rerank(lines, nbest, las_rescore=False, rnnt_score_scale=1.0, las_fw_score_scale=0.3,
           las_bw_score_scale=0.7):
    """Yields the winning hypothesis (a string) of every complete group of `nbest` lines."""
    group = []
    for line in lines:
        hyp, score = score_line(line.split(), las_rescore, rnnt_score_scale, las_fw_score_scale,
                                las_bw_score_scale)
        norm = 0.001 if len(hyp) == 0 else len(hyp)
        group.append((-score / norm, hyp))
        if len(group) == nbest:
            yield min(group, key=lambda x: x[0])[1]     # min keeps the first of equal costs, like a stable sort
            group = []


def main(argv=None):
    ap = argparse.ArgumentParser(description="pick the best hypothesis of every n-best group")
    ap.add_argument("in_hyp")
    ap.add_argument("out_hyp")
    ap.add_argument("--nbest", type=int)
    ap.add_argument("--las_rescore", action="store_true")
    ap.add_argument("--rnnt_score_scale", type=float, default=1.0)
    ap.add_argument("--las_fw_score_scale", type=float, default=0.3)
    ap.add_argument("--las_bw_score_scale", type=float, default=0.7)
    args, _ = ap.parse_known_args(argv)
    with open(args.in_hyp, "r", encoding="utf-8") as fi, open(args.out_hyp, "w", encoding="utf-8") as fo:
        for hyp in rerank(fi, args.nbest, args.las_rescore, args.rnnt_score_scale, args.las_fw_score_scale,
                          args.las_bw_score_scale):
            fo.write("{}\n".format(" ".join(hyp)))


if __name__ == "__main__":
    main()
