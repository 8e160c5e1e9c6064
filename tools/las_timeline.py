#!/usr/bin/env python
"""The LAS rescoring of the LAST decode batch in a rocprofv3 kernel trace (rocpd .db): device span, busy and idle time, the
large gaps with their neighbours (= host work the device waits for) and the busy time by kernel.
    python tools/las_timeline.py NAME_results.db"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = con.execute("select %s, start, end from kernels order by start" % name).fetchall()
    enc = [i for i, r in enumerate(rows) if "blstm_layer_kernel" in r[0]]
    first = enc[-4]                                   # 2 layers x 2 rescorers per batch
    while first > 0 and rows[first][1] - rows[first - 1][2] < 200e3 and "beam_" not in rows[first - 1][0]:
        first -= 1                                    # (the encoder's input products in front of the first layer)
    win = rows[first:]
    t0 = win[0][1]
    busy, by = 0, {}
    gaps = []
    end = win[0][1]
    for n, s, e in win:
        if s > end:
            gaps.append((s - end, (end - t0) / 1e6, prev, n[:60]))
        busy += max(0, e - max(s, end))
        by[n[:70]] = by.get(n[:70], 0) + (e - s)
        end, prev = max(end, e), n[:60]
    span = end - t0
    print("LAS rescoring of the last batch: %d kernels, span %.2f ms, busy %.2f ms, idle %.2f ms" % (
        len(win), span / 1e6, busy / 1e6, (span - busy) / 1e6))
    print("gaps > 50 us (at ms: before -> after):")
    for g, at, a, b in gaps:
        if g > 50e3:
            print("  %7.2f ms at %6.2f  %s -> %s" % (g / 1e6, at, a, b))
    small = sum(g for g, _, _, _ in gaps if g <= 50e3)
    print("gaps <= 50 us: %d, %.2f ms in all" % (sum(1 for g in gaps if g[0] <= 50e3), small / 1e6))
    print("busy time by kernel:")
    for n, t in sorted(by.items(), key=lambda kv: -kv[1])[:22]:
        print("  %7.2f ms  %s" % (t / 1e6, n))


if __name__ == "__main__":
    main()
