#!/usr/bin/env python
"""Launch-chain anatomy of the beam-search step from a rocprofv3 kernel trace: for every kernel of the step chain the
number of launches, the average duration and the average idle gap in front of it (gaps > 200 us are host waits, not chain
latency, and are left out).    python tools/step_chain.py NAME_results.db"""
import sqlite3
import sys

STEP = ("dgemm_kernel", "dgemm_sk_kernel", "ln_fwd_kernel", "dstep_", "dfc2_", "beam_partials", "fst_advance", "lstm_cell")
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = con.execute("select %s, start, end from kernels order by start" % name).fetchall()
agg = {}
prev_end = None
for n, s, e in rows:
    if any(k in n for k in STEP):
        key = n.replace("(anonymous namespace)::", "").replace("void ", "")[:56]
        a = agg.setdefault(key, [0, 0.0, 0, 0.0])
        a[0] += 1
        a[1] += e - s
        if prev_end is not None and 0 <= s - prev_end < 200000:
            a[2] += 1
            a[3] += s - prev_end
    prev_end = e if prev_end is None else max(prev_end, e)
tot_d = tot_g = 0.0
n_steps = max(agg.get(k, [0])[0] for k in agg if "beam_partials" in k) if any("beam_partials" in k for k in agg) else 1
print("%-50s %8s %10s %10s %12s" % ("kernel", "launches", "avg us", "gap us", "us per step"))
for k, (c, d, gc, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    print("%-50s %8d %10.2f %10.2f %12.1f" % (k, c, d / c / 1e3, g / max(gc, 1) / 1e3, (d + g) / n_steps / 1e3))
    tot_d += d
    tot_g += g
print("search steps %d: kernels %.1f us + gaps %.1f us per step" % (n_steps, tot_d / n_steps / 1e3, tot_g / n_steps / 1e3))
