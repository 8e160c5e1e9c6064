#!/usr/bin/env python
"""Where pika_beam_advance_* spends its time: time stamps of utterance 0's first wave (s_memrealtime, 10 ns ticks), averaged
over the launches of a decode.  Needs a profiling build:  PIKA_HIPCC_EXTRA=-DPIKA_ADV_TRACE python -m pika_amd.build --force
    python tools/adv_trace.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--workload", "decode", "--batch", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
import runpy
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
from pika_amd import _lib
buf = (ctypes.c_ulonglong * 16)()
h = _lib.lib()
h.pika_debug_adv_trace.argtypes = [ctypes.c_void_p]
assert h.pika_debug_adv_trace(buf) == 0
n = max(int(buf[0]), 1)
names = {1: "stop flag read", 2: "slot state + hypotheses in LDS", 3: "duplicate check (row 0)", 4: "row statistics merged",
         5: "survivors of the thresholded row pass", 6: "K best of the survivors", 7: "all rows done (barrier)",
         8: "merge + bookkeeping", 9: "done flags / step counter"}
names.update({10: "  rank sort of the K*K row winners", 11: "  bookkeeping on K lanes", 12: "  finished list"})
prev = 0.0
print("launches %d" % n)
for k in (1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 8, 9):
    t = buf[k] / n / 100.0
    print("%-48s at %7.2f us  (+%.2f)" % (names[k], t, t - prev))
    prev = t
