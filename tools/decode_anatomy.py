#!/usr/bin/env python
"""The LAST decode batch of a rocprofv3 kernel trace (rocpd .db) of `bench.py --workload decode`: the encoder pass + joint
halves + search set-up (everything between the previous batch's last search kernel and this batch's first dstep_prep) by
kernel, and the search loop's span.      python tools/decode_anatomy.py NAME_results.db"""
import sqlite3
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]


con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = con.execute("select %s, start, end from kernels order by start" % name).fetchall()
prep = [i for i, r in enumerate(rows) if "dstep_prep" in r[0]]
runs, cur = [], [prep[0]]
for i in prep[1:]:
    if rows[i][1] - rows[cur[-1]][1] > 3e6:
        runs.append(cur); cur = [i]
    else:
        cur.append(i)
runs.append(cur)
a1 = runs[-2][-1]
while a1 + 1 < len(rows) and rows[a1 + 1][1] - rows[a1][2] < 1e6 and ("beam_" in rows[a1 + 1][0] or "dfc2" in rows[a1 + 1][0] or "dgemm" in rows[a1 + 1][0] or "dstep" in rows[a1 + 1][0]):
    a1 += 1
b0 = runs[-1][0]
seg = rows[a1 + 1:b0]
t0, end, busy, by = seg[0][1], seg[0][1], 0, {}
gaps = []
prev = None
for n, s, e in seg:
    if s > end and prev:
        gaps.append((s - end, (end - t0) / 1e6, prev, short(n)))
    busy += max(0, e - max(s, end))
    k = short(n)
    by.setdefault(k, [0, 0]); by[k][0] += e - s; by[k][1] += 1
    end, prev = max(end, e), k
print("between two searches (n-best read-out of batch k, then encoder pass + joint halves + set-up of batch k + 1): %d launches, "
      "span %.2f ms, busy %.2f ms" % (len(seg), (end - t0) / 1e6, busy / 1e6))
for g, at, a, b in sorted(gaps, reverse=True)[:8]:
    print("    gap %6.2f ms at %6.2f ms: %s -> %s" % (g / 1e6, at, a, b))
for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:24]:
    print("    %7.3f ms  %5d x  %s" % (t / 1e6, c, k))
if "--seq" in sys.argv:     # the launches between the two searches in order: start (us), duration, gap, kernel
    end = seg[0][1]
    for n, s_, e in seg:
        print("  %9.1f %8.1f %7.1f  %s" % ((s_ - t0) / 1e3, (e - s_) / 1e3, max(0, s_ - end) / 1e3, short(n)))
        end = max(end, e)
last = rows[runs[-1][0]:]
print("search loop of the last batch: %.2f ms from its first dstep_prep to the last kernel of the trace region" % (
    (last[-1][2] - last[0][1]) / 1e6))
