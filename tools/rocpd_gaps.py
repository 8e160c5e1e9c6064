#!/usr/bin/env python
"""Where the device idles inside a traced run: gaps between consecutive kernels (all queues merged: a gap is time in
which NO kernel runs), summed by the kernel that FOLLOWS the gap, plus the largest single gaps.
    python tools/rocpd_gaps.py NAME_results.db [--skip-first N_KERNELS] [--min-us 1]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 0
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = con.execute("select %s, start, end from kernels order by start" % name).fetchall()[skip:]
    busy_end = rows[0][2]
    gaps = {}
    big = []
    tot_gap = 0
    hist = [0, 0, 0, 0, 0]   # <2us, 2-5, 5-20, 20-100, >100
    for i in range(1, len(rows)):
        n, s, e = rows[i]
        g = s - busy_end
        if g > 0:
            tot_gap += g
            a = gaps.setdefault(n[:70], [0, 0])
            a[0] += 1
            a[1] += g
            big.append((g, rows[i - 1][0][:50], n[:50]))
            us = g / 1e3
            hist[0 if us < 2 else 1 if us < 5 else 2 if us < 20 else 3 if us < 100 else 4] += g
        busy_end = max(busy_end, e)
    span = rows[-1][2] - rows[0][1]
    print("kernels %d  span %.2f ms  idle %.2f ms (%.1f%%)" % (len(rows), span / 1e6, tot_gap / 1e6, 100.0 * tot_gap / span))
    print("idle by gap size  <2us %.2f  2-5us %.2f  5-20us %.2f  20-100us %.2f  >100us %.2f ms" % tuple(h / 1e6 for h in hist))
    print("\nidle by the kernel that follows the gap:")
    for n, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %8.3f ms  %5d gaps  %6.1f us avg  %s" % (g / 1e6, c, g / c / 1e3, n))
    print("\nlargest gaps (us): before <- after")
    for g, a, b in sorted(big, reverse=True)[:25]:
        print("  %9.1f  %s -> %s" % (g / 1e3, a, b))


if __name__ == "__main__":
    main()
