#!/usr/bin/env python
"""The ordered launch sequence of the LAST training step in a rocprofv3 kernel trace (rocpd .db) of
`bench.py --workload train_step`: every launch between two consecutive sgd_kernel dispatches with its start offset,
duration, gap to the previous launch's end and grid -- what to fold into what.
    python tools/step_sequence.py NAME_results.db [> profiles/rN_train_step_sequence.txt]"""
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("_ZN"):
        import re
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
        if m:
            k = int(m.group(1))
            i = len("_ZN12_GLOBAL__N_1") + len(m.group(1))
            return n[i:i + k] + "<" + n[i + k:i + k + 24] + ">"
    return n.split("(")[0][:70]


con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
gx = "grid_x" if "grid_x" in cols else None
rows = con.execute("select %s, start, end%s from kernels order by start" % (name, ", " + gx if gx else "")).fetchall()
sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r[0]]
step = rows[sgd[-2] + 1: sgd[-1] + 1]
t0, end = step[0][1], step[0][1]
small_n = small_t = 0
print("# %d launches, %.3f ms from the first launch behind the previous sgd_kernel to the end of this step's" % (
    len(step), (step[-1][2] - t0) / 1e6))
print("#   at(us)   dur(us)  gap(us)  kernel  [grid]")
for r in step:
    d = (r[2] - r[1]) / 1e3
    if d < 30:
        small_n += 1
        small_t += d
    print("%9.1f %8.1f %7.1f  %s  [%s]" % ((r[1] - t0) / 1e3, d, max(0.0, (r[1] - end) / 1e3), short(r[0]), r[3] if gx else "?"))
    end = max(end, r[2])
print("# launches shorter than 30 us: %d, %.2f ms in all" % (small_n, small_t / 1e3))
