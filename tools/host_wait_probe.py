"""How much CPU does the training thread burn while it waits for the device in `loss.item()`?  (The reference loop reads
the loss every step, train_transducer_bmuf_otfaug.py:129-130.)  Thread CPU time vs wall time of `.item()` behind ~40 ms of
queued device work, with the default device flags and after hipSetDeviceFlags(hipDeviceScheduleBlockingSync)."""
import ctypes
import sys
import time

import torch


def measure(tag, n=10):
    a = torch.randn(8192, 8192, device="cuda")
    s = torch.zeros((), device="cuda")
    torch.cuda.synchronize()
    cpu = wall = 0.0
    for _ in range(n):
        for _ in range(24):
            b = a @ a
        s = b[0, 0] + 1.0
        t0, c0 = time.perf_counter(), time.thread_time()
        s.item()
        wall += time.perf_counter() - t0
        cpu += time.thread_time() - c0
    print("%-28s wall %.2f ms  thread CPU %.2f ms per .item()" % (tag, wall / n * 1e3, cpu / n * 1e3), flush=True)


if __name__ == "__main__":
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    if len(sys.argv) > 1 and sys.argv[1] == "blocking":
        hip = ctypes.CDLL("libamdhip64.so")
        rc = hip.hipSetDeviceFlags(ctypes.c_uint(4))          # hipDeviceScheduleBlockingSync
        print("hipSetDeviceFlags(BlockingSync) ->", rc)
        measure("blocking-sync device flag")
    else:
        measure("default device flags")
