"""Hardware A/B of the register-staged GEMM on the SMALL products of the train step (prediction net, joint halves:
1632 / 7680 rows), which run on 8..28 workgroups today.  One subprocess per environment override.
    python tools/small_gemm_bench.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NN = [("qkv/out  1632x512x512", 1632, 512, 512), ("ffn1     1632x2048x512", 1632, 2048, 512),
      ("ffn2     1632x512x2048", 1632, 512, 2048), ("conv     1632x512x2560", 1632, 512, 2560),
      ("lin_out  1632x1024x512", 1632, 1024, 512), ("p1/pg    1632x1024x1024", 1632, 1024, 1024),
      ("e1/eg    7680x1024x1024", 7680, 1024, 1024)]
TT = [("dW 512x512  r1632", 512, 512, 1632), ("dW 2048x512 r1632", 2048, 512, 1632), ("dW 512x2048 r1632", 512, 2048, 1632),
      ("dW 512x2560 r1632", 512, 2560, 1632), ("dW 1024x1024 r1632", 1024, 1024, 1632), ("dW 1024x1024 r7680", 1024, 1024, 7680)]


def worker():
    import torch
    sys.path.insert(0, ROOT)
    from pika_amd import gemm as G
    dev = torch.device("cuda:0")
    tag = os.environ.get("TAG", "")

    def timeit(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    for name, M, N, K in NN:
        a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        out = torch.empty(M, N, device=dev)
        us = timeit(lambda: G.gemm_nt(a, b, out=out, precision="bf16"))
        print("%-28s NN %-26s %8.1f us" % (tag, name, us), flush=True)
    for name, M, N, R in TT:
        at, bt = torch.randn(R, M, device=dev), torch.randn(R, N, device=dev)
        out = torch.empty(M, N, device=dev)
        a_op, b_op = G.matrix_t(at)[0], G.matrix_t(bt)[0]
        us = timeit(lambda: G.launch(a_op, b_op, out, N, M, N, R, precision="bf16"))
        print("%-28s TT %-26s %8.1f us" % (tag, name, us), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        runs = [{}, {"PIKA_GEMM_CFG": "1"}, {"PIKA_GEMM_CFG": "0"}, {"PIKA_GEMM_CFG_T": "1"},
                {"PIKA_GEMM_SPLIT_TARGET": "64", "PIKA_GEMM_SPLIT_MIN_NK": "8", "PIKA_GEMM_SPLIT_MIN_PER": "2"},
                {"PIKA_GEMM_SPLIT_TARGET": "128", "PIKA_GEMM_SPLIT_MIN_NK": "8", "PIKA_GEMM_SPLIT_MIN_PER": "2"},
                {"PIKA_GEMM_CFG": "1", "PIKA_GEMM_CFG_T": "1", "PIKA_GEMM_SPLIT_TARGET": "128", "PIKA_GEMM_SPLIT_MIN_NK": "8",
                 "PIKA_GEMM_SPLIT_MIN_PER": "2"},
                {"PIKA_GEMM_CFG": "1", "PIKA_GEMM_CFG_T": "1", "PIKA_GEMM_SPLIT_TARGET": "256", "PIKA_GEMM_SPLIT_MIN_NK": "8",
                 "PIKA_GEMM_SPLIT_MIN_PER": "2"}]
        for r in runs:
            env = dict(os.environ, TAG=",".join("%s=%s" % (k[10:], v) for k, v in r.items()) or "default", **r)
            subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=env, check=False)
