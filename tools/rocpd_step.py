#!/usr/bin/env python
"""Timeline of ONE steady-state beam-search step from a rocprofv3 rocpd database: the launches between two
consecutive dstep_prep_kernel dispatches in the middle of the run (start offset, duration, short name, grid)."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name,start,end,grid_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "dstep_prep" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
i0, i1 = idx[k], idx[k + 1]
t0 = rows[i0][1]
busy = 0
for r in rows[i0:i1]:
    n = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    busy += r[2] - r[1]
    print("%8.1f +%7.1f us  %-28s grid %d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, n[:28], r[3]))
print("# step span %.1f us, busy %.1f us, %d launches" % ((rows[i1][1] - t0) / 1e3, busy / 1e3, i1 - i0))
