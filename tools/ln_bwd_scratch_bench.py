"""pika_layer_norm_bwd at the encoder's shapes, float atomics onto dgamma / dbeta (partials NULL: 2 memsets + 1 launch) against
the per-workgroup partials + fixed-order second launch.  GPU box: python tools/ln_bwd_scratch_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pika_amd  # noqa: E402,F401
import torch  # noqa: E402
from pika_amd import _lib  # noqa: E402
from pika_amd import gemm as G  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
for rows, C in ((31808, 512), (31808, 1024), (7680, 512), (1632, 512)):
    x = torch.randn(rows, C, device=dev)
    w = torch.randn(C, device=dev)
    dy = torch.randn(rows, C, device=dev).bfloat16()
    mean, rstd = x.mean(1).contiguous(), (x.var(1, unbiased=False) + 1e-6).rsqrt().contiguous()
    dx, dg, db = torch.empty_like(x), torch.empty(C, device=dev), torch.empty(C, device=dev)
    part = torch.empty(int(lib.pika_layer_norm_bwd_partial_floats(rows, C)), device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def call(p):
        _lib.check(lib.pika_layer_norm_bwd(dy.data_ptr(), G.PIKA_BF16, x.data_ptr(), rows, C, w.data_ptr(), mean.data_ptr(),
                                           rstd.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), p, None, st), "ln_bwd")
    out = []
    for p in (None, part.data_ptr()):
        for _ in range(5):
            call(p)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call(p)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 50 * 1e3)
    print("%6d x %4d: atomics %.1f us, partials %.1f us" % (rows, C, out[0], out[1]), flush=True)
