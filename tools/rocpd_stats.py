#!/usr/bin/env python
"""Per-kernel summary (calls, total / average / min / max duration) of a rocprofv3 `*_results.db` (rocpd sqlite
output of ROCm 7: `rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`), as CSV on stdout.
    python tools/rocpd_stats.py gpurun_out/x/prof/NAME_results.db [--top N] [--since-last KERNEL_SUBSTR]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 60
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = con.execute("select %s, start, end from kernels order by start" % name).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('"%s",%d,%d,%.0f,%.2f,%d,%d' % (n[:150], a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3]))
    if rows:
        print("# kernels %d, busy %.3f ms, span %.3f ms" % (len(rows), tot / 1e6, (rows[-1][2] - rows[0][1]) / 1e6))


if __name__ == "__main__":
    main()
