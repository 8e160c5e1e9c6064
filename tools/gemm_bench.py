"""Hardware A/B of the GEMM tile configurations on the shapes of the config-2 train step.
Run on the GPU box:  python tools/gemm_bench.py   (spawns one process per PIKA_GEMM_CFG)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [  # name, M, N, K, dtype A, dtype B
    ("tdnn_fwd  f32xf32", 32000, 1024, 3072, "f32", "f32"),
    ("ffn_fwd   f32xf32", 32000, 4096, 1024, "f32", "f32"),
    ("fc2_fwd   bf16xf32", 391680, 5000, 1024, "bf16", "f32"),
    ("fc2_dh    f32xbf16", 391680, 1024, 5000, "f32", "bf16"),
    ("fc2_dW    bf16xbf16", 5000, 1024, 391680, "bf16", "bf16"),
    ("tdnn_dW   bf16xbf16 splitK", 1024, 3072, 32000, "bf16", "bf16"),
    ("tdnn_dX   f32xbf16", 32000, 3072, 1024, "f32", "bf16"),
]


def worker():
    import torch
    sys.path.insert(0, ROOT)
    from pika_amd import gemm as G
    dev = torch.device("cuda:0")
    for name, M, N, K, da, db in SHAPES:
        a = torch.randn(M, K, device=dev)
        b = torch.randn(N, K, device=dev)
        if da == "bf16":
            a = a.bfloat16()
        if db == "bf16":
            b = b.bfloat16()
        out = torch.empty(M, N, device=dev)
        for _ in range(2):
            G.gemm_nt(a, b, out=out, precision="bf16")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 5
        for _ in range(reps):
            G.gemm_nt(a, b, out=out, precision="bf16")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("cfg=%s  %-30s %9.3f ms  %7.1f TFLOP/s" % (os.environ.get("PIKA_GEMM_CFG"), name, ms,
                                                         2.0 * M * N * K / ms / 1e9), flush=True)
        del a, b, out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        for cfg in (sys.argv[1:] or ["0", "1", "2", "3"]):
            env = dict(os.environ, PIKA_GEMM_CFG=cfg)
            subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=env, check=False)
