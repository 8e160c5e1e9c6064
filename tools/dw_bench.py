"""Hardware A/B of the split-K target for the weight-gradient (transposed-operand) GEMMs.
Run on the GPU box:  python tools/dw_bench.py [targets...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("tdnn_dW", 1024, 3072, 31616), ("proj_dW", 1024, 1024, 31616), ("ffn1_dW", 4096, 1024, 31616),
          ("ffn2_dW", 1024, 4096, 31616), ("late_dW", 1024, 1024, 7680), ("joint_dW2", 5056, 1024, 391680)]


def worker():
    import torch
    sys.path.insert(0, ROOT)
    from pika_amd import gemm as G
    dev = torch.device("cuda:0")
    only = os.environ.get("DW_BENCH_ONLY")          # e.g. joint_dW2: one shape (for --pmc passes)
    for name, N, Ka, M in SHAPES:
        if only and name != only:
            continue
        dy = torch.randn(M, N, device=dev).bfloat16()
        x = torch.randn(M, Ka, device=dev).bfloat16()
        out = torch.empty(N, Ka, device=dev)

        def run():
            a, b = G.matrix(dy)[0], G.matrix(x)[0]
            a.trans = b.trans = 1
            G.launch(a, b, out, Ka, N, Ka, M)
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("target=%s  %-10s %8.3f ms  %7.1f TFLOP/s" % (os.environ.get("PIKA_GEMM_SPLIT_TARGET"), name, ms,
                                                            2.0 * M * N * Ka / ms / 1e9), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        var = os.environ.get("DW_BENCH_VAR", "PIKA_GEMM_SPLIT_TARGET")
        for t in (sys.argv[1:] or ["128", "192", "256", "320", "384", "512", "768", "1024"]):
            subprocess.run([sys.executable, os.path.abspath(__file__), "worker"],
                           env=dict(os.environ, **{var: t, "PIKA_GEMM_SPLIT_TARGET": t}), check=False)
