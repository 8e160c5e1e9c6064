#!/usr/bin/env python
"""Anatomy of the LAST MBR step in a rocprofv3 kernel trace (rocpd .db) of `bench.py --workload mbr_step`: the step is
cut at its marker kernels into (a) the decoder's encoder pass + joint halves, (b) the N-best search loop (first to last
dstep_prep_kernel .. beam kernels), (c) the n-best read-out + risk terms (host) and the TRAINING half up to the optimizer's
sgd_kernel.  Per part: span, device-busy time, idle time, the gaps > 50 us with their neighbours, busy time by kernel.
    python tools/mbr_anatomy.py NAME_results.db"""
import sqlite3
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]


def part(rows, label):
    if not rows:
        print("%s: empty" % label)
        return
    t0, end, busy, by, gaps, prev = rows[0][1], rows[0][1], 0, {}, [], None
    for n, s, e in rows:
        if s > end and prev is not None:
            gaps.append((s - end, (end - t0) / 1e6, prev, short(n)))
        busy += max(0, e - max(s, end))
        by[short(n)] = by.get(short(n), [0, 0])
        by[short(n)][0] += e - s
        by[short(n)][1] += 1
        end, prev = max(end, e), short(n)
    span = end - t0
    print("%s: %d launches, span %.2f ms, busy %.2f ms, idle %.2f ms" % (label, len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6))
    big = [g for g in gaps if g[0] > 50e3]
    for g, at, a, b in big[:12]:
        print("    gap %6.2f ms at %6.2f ms: %s -> %s" % (g / 1e6, at, a, b))
    print("    gaps <= 50 us: %d, %.2f ms in all" % (len(gaps) - len(big), sum(g[0] for g in gaps if g[0] <= 50e3) / 1e6))
    for n, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:16]:
        print("    %7.3f ms  %5d x  %s" % (t / 1e6, c, n))


def main():
    con = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = con.execute("select %s, start, end from kernels order by start" % name).fetchall()
    is_search = lambda n: ("dstep_" in n or "beam_" in n or "dfc2" in n or "dgemm_sk" in n or "fst_advance" in n)
    prep = [i for i, r in enumerate(rows) if "dstep_prep" in r[0]]
    # search loops = runs of dstep_prep launches less than 3 ms apart
    runs, cur = [], [prep[0]]
    for i in prep[1:]:
        if rows[i][1] - rows[cur[-1]][1] > 3e6:
            runs.append(cur)
            cur = [i]
        else:
            cur.append(i)
    runs.append(cur)
    assert len(runs) >= 2, "need two N-best searches in the trace"
    a0 = runs[-2][0]
    a1 = max(i for i in range(runs[-2][-1], runs[-1][0]) if is_search(rows[i][0]) and rows[i][1] - rows[runs[-2][-1]][1] < 3e6)
    b0 = runs[-1][0]
    print("one MBR step = N-best search k, then training half k, then the decoder's encoder pass of step k + 1: %.2f ms from the "
          "first launch of search k to the first launch of search k + 1" % ((rows[b0][1] - rows[a0][1]) / 1e6))
    part(rows[a0:a1 + 1], "(b) N-best search loop")
    rest = rows[a1 + 1:b0]
    print("    host time between the search's last kernel and the next launch: %.2f ms" % ((rest[0][1] - rows[a1][2]) / 1e6))
    # the training half ends with the optimizer (the last multi-tensor / sgd launch before the next decode's first GEMM)
    opt = [i for i, r in enumerate(rest) if "multi_tensor" in r[0] or "sgd_kernel" in r[0] or "scale_kernel" in r[0]]
    cut = (opt[-1] + 1) if opt else len(rest)
    part(rest[:cut], "(c) n-best read-out, risk terms, TRAINING half, clip + SGD")
    part(rest[cut:], "(a) next step's decoder encoder pass + joint halves + search set-up")


if __name__ == "__main__":
    main()
