"""Time stamps inside gemm_nt_kernel (a -DPIKA_NT_TRACE build: PIKA_HIPCC_EXTRA=-DPIKA_NT_TRACE python -m pika_amd.build --force):
workgroup 0's first lane at kernel entry, after the loaders' set-up, after the first tile is staged, and per K-step after the
requests are issued / after the MFMAs / after the staging (= the wait for the requests) / after the barrier.  100 MHz counter."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pika_amd  # noqa: E402,F401
import torch  # noqa: E402
from pika_amd import _lib  # noqa: E402
from pika_amd import gemm as G  # noqa: E402

dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(ROOT, "pika_amd", "libpika_amd.so"))
for name, M, N, K in (("qkv 3 terms 1632x512x1536", 1632, 512, 1536), ("1632x2048x1536", 1632, 2048, 1536), ("8192x512x1536", 8192, 512, 1536)):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev)
    for _ in range(3):
        G.gemm_nt(a, b, out=out, precision="bf16")
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    assert lib.pika_debug_nt_trace(buf) == 0
    t = [int(v) for v in buf]
    us = lambda i, j: (t[j] - t[i]) / 100.0
    print("%s: entry->setup %.2f us, first tile requested+staged %.2f, barrier %.2f" % (name, us(0, 1), us(1, 2), us(2, 3)))
    for kb in range(6):
        base = 4 + 4 * kb
        prev = 3 if kb == 0 else base - 1
        print("   step %d: requests issued %.2f  mfma %.2f  staging (wait) %.2f  barrier %.2f   = %.2f us" % (
            kb, us(prev, base), us(base, base + 1), us(base + 1, base + 2), us(base + 2, base + 3), us(prev, base + 3)))
    print("   K loop in all (%d steps): %.2f us; entry -> loop end %.2f us" % (K // 64, us(3, 62), us(0, 62)))
