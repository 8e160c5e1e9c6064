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
    sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r[0]]
    assert len(sgd) >= 2, "need two optimizer steps in the trace"
    step = rows[sgd[-2] + 1: sgd[-1] + 1]
    prep = [i for i, r in enumerate(step) if "dstep_prep" in r[0]]
    beam = [i for i, r in enumerate(step) if "beam_" in r[0] or "dstep_" in r[0] or "dfc2" in r[0]]
    a, b = prep[0], beam[-1]
    print("last MBR step: %d launches, %.2f ms from the first launch after the previous optimizer step to the end of sgd_kernel"
          % (len(step), (step[-1][2] - step[0][1]) / 1e6))
    part(step[:a], "(a) decoder's encoder pass + joint halves")
    part(step[a:b + 1], "(b) N-best search loop")
    print("    host time between the search's last kernel and the training half's first: %.2f ms" % (
        (step[b + 1][1] - step[b][2]) / 1e6))
    part(step[b + 1:], "(c) training half (incl. clip + SGD)")


if __name__ == "__main__":
    main()
