#!/usr/bin/env python
"""Where a tile of pika_dgemm's split-reduction kernel spends its time: time stamps of workgroup 0's first wave (s_memrealtime,
10 ns ticks), averaged over the launches of a decode, per kernel variant.  Needs a profiling build:
    PIKA_HIPCC_EXTRA=-DPIKA_SK_TRACE python -m pika_amd.build --force;  python tools/sk_trace.py"""
import ctypes
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--workload", "decode", "--batch", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
from pika_amd import _lib
buf = (ctypes.c_ulonglong * 64)()
h = _lib.lib()
h.pika_debug_sk_trace.argtypes = [ctypes.c_void_p]
assert h.pika_debug_sk_trace(buf) == 0
names = ["", "row count read (m_dev)", "tile set up (indices, bases)", "operands requested", "operands arrived (+ LayerNorm statistics) + MFMAs",
         "partial tiles merged (barrier)", "epilogue stored"]
for v in range(8):
    n = int(buf[v * 8])
    if not n:
        continue
    print("variant LN=%d KW=%d PIPE=%d: %d launches" % (v >> 2, 8 if v & 2 else 4, v & 1, n))
    prev = 0.0
    for k in range(1, 7):
        t = buf[v * 8 + k] / n / 100.0
        print("   %-44s at %6.2f us (+%.2f)" % (names[k], t, t - prev))
        prev = t
