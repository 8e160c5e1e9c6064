"""`editdistance.eval(a, b)` (requirements.txt:1; used by the MBR trainer,
trainer/train_transducer_mbr_bmuf_otfaug.py:24,188): Levenshtein distance of two sequences."""


def eval(a, b):  # noqa: A001  (name fixed by the package being replaced)
    a, b = list(a), list(b)
    if a and b and all(type(x) is int for x in a) and all(type(x) is int for x in b):
        try:        # integer label sequences (what the MBR trainer passes): the library's host routine
            from pika_amd.mbr import edit_distances
            return edit_distances([(a, b)])[0]
        except (OverflowError, RuntimeError, OSError):
            pass
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


distance = eval
