"""BatchNorm statistics / backward at the encoder's shape under PIKA_BN_RPB (rows per reduction workgroup)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    import torch
    sys.path.insert(0, ROOT)
    from pika_amd import _lib
    dev = torch.device("cuda:0")
    M, C = 31808, 1024
    x = torch.randn(M, C, device=dev); dy = torch.randn(M, C, device=dev).bfloat16()
    stats = torch.empty(2 * C, dtype=torch.float64, device=dev); sums = torch.empty_like(stats)
    mean = torch.zeros(C, device=dev); rstd = torch.ones(C, device=dev); g = torch.ones(C, device=dev)
    dx = torch.empty(M, C, dtype=torch.bfloat16, device=dev); dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
    lib = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
    def fwd(): _lib.check(lib.pika_bn_stats(x.data_ptr(), M, C, stats.data_ptr(), None, st), "stats")
    def bwd(): _lib.check(lib.pika_bn_backward(dy.data_ptr(), 1, x.data_ptr(), M, C, g.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               sums.data_ptr(), dx.data_ptr(), 1, dg.data_ptr(), db.data_ptr(), 1, None, st), "bwd")
    for name, fn in (("stats", fwd), ("backward (reduce + apply + param grads)", bwd)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print("rpb=%s  %-40s %.1f us" % (os.environ.get("PIKA_BN_RPB"), name, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
else:
    for r in ("512", "256", "128", "64", "32"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=dict(os.environ, PIKA_BN_RPB=r))
