"""Hardware A/B of the persistent direct-to-LDS GEMM (PIKA_GEMM_PP_WGS: 0 = one workgroup per tile).
Run on the GPU box:  python tools/pp_bench.py [wgs...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (name, M, N, K, bf16 output)
SHAPES = [("joint_logits", 391680, 5000, 1024, False), ("joint_dh", 391680, 1024, 5056, True),
          ("ffn1", 31616, 4096, 1024, True), ("ffn2", 31616, 1024, 4096, False), ("proj", 31616, 1024, 1024, False),
          ("tdnn", 31616, 1024, 3072, True), ("late", 7680, 1024, 1024, False), ("square", 8192, 8192, 8192, False)]


def worker():
    import torch
    sys.path.insert(0, ROOT)
    from pika_amd import gemm as G
    dev = torch.device("cuda:0")
    for name, M, N, K, o16 in SHAPES:
        a = torch.randn(M, K, device=dev).bfloat16()
        b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if o16 else torch.float32)

        def run():
            G.launch(G.matrix(a)[0], G.matrix(b)[0], out, N, M, N, K, bias=bias)
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5 if M > 100000 else 20
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        ref = (a[:256].float() @ b.float().t() + bias)
        err = (out[:256].float() - ref).abs().max().item() / ref.abs().max().item()
        print("wgs=%-4s %-13s %8.3f ms  %7.1f TFLOP/s   relerr %.1e" % (os.environ.get("PIKA_GEMM_PP_WGS"), name, ms,
                                                                     2.0 * M * N * K / ms / 1e9, err), flush=True)
        del a, b, out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        for t in (sys.argv[1:] or ["0", "256"]):
            print("--- PIKA_GEMM_PP_WGS=%s" % t, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=dict(os.environ, PIKA_GEMM_PP_WGS=t),
                           check=False)
