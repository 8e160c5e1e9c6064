#!/usr/bin/env python
"""Who issues the device copies?  Every `__amd_rocclr_copyBuffer` dispatch of a rocprofv3 kernel trace (rocpd .db) binned by
(a) the phase of the decode batch it falls in -- encoder pass / search loop / n-best read-out / LAS rescoring, delimited by
marker kernels -- and (b) the nearest preceding non-copy kernel; plus the memory-copy records themselves (direction, bytes)
when the trace was taken with --memory-copy-trace.
    python tools/copy_census.py NAME_results.db"""
import collections
import sqlite3
import sys

PHASES = [  # (label, substring of a kernel that can only belong to the phase)
    ("search", "beam_partials"), ("search", "dstep_"), ("search", "dfc2_"), ("search", "dgemm_sk"),
    ("las", "las_"), ("las", "blstm_layer"), ("las", "dgemm_wide"), ("las", "lstm_cell"), ("las", "dgemm_kernel"),
    ("encoder", "attn_"), ("encoder", "gemm_pp"), ("encoder", "bn_"), ("encoder", "ln_"), ("encoder", "split_terms"),
    ("frontend", "fbank"), ("frontend", "splice"),
]


def phase_of(name):
    for lab, sub in PHASES:
        if sub in name:
            return lab
    return None


def main():
    con = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = con.execute("select %s, start, end from kernels order by start" % name).fetchall()
    by_phase, by_prev, by_next = collections.Counter(), collections.Counter(), collections.Counter()
    t_phase = collections.Counter()
    cur, prev = "start-up", "(none)"
    pending = []
    n_copy = 0
    for n, s, e in rows:
        if "copyBuffer" in n:
            n_copy += 1
            by_phase[cur] += 1
            t_phase[cur] += e - s
            by_prev[(cur, prev[:70])] += 1
            pending.append(cur)
            continue
        for c in pending:
            by_next[(c, n[:70])] += 1
        pending = []
        prev = n
        p = phase_of(n)
        if p is not None:
            cur = p
    print("copyBuffer dispatches: %d" % n_copy)
    print("by phase (the phase of the last marker kernel in front of the copy):")
    for k, v in by_phase.most_common():
        print("  %-10s %6d   %.2f ms" % (k, v, t_phase[k] / 1e6))
    print("by (phase, preceding kernel), top 25:")
    for (ph, k), v in by_prev.most_common(25):
        print("  %6d  %-9s after  %s" % (v, ph, k))
    print("by (phase, following kernel), top 25:")
    for (ph, k), v in by_next.most_common(25):
        print("  %6d  %-9s before %s" % (v, ph, k))
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    mc = [t for t in tabs if t == "memory_copies"] or [t for t in tabs if "memory_cop" in t]
    if mc:
        c2 = [r[1] for r in con.execute("pragma table_info(%s)" % mc[0])]
        nm = "name" if "name" in c2 else None
        sz = "size" if "size" in c2 else None
        if nm and sz:
            agg = collections.Counter()
            byts = collections.Counter()
            small = collections.Counter()
            for n, b in con.execute("select %s, %s from %s" % (nm, sz, mc[0])):
                agg[n] += 1
                byts[n] += b or 0
                if (b or 0) <= 256:
                    small[n] += 1
            print("memory-copy records (%s):" % mc[0])
            for k, v in agg.most_common():
                print("  %-40s %7d copies  %10.2f MB  (%d of them <= 256 B)" % (k, v, byts[k] / 1e6, small[k]))


if __name__ == "__main__":
    main()
