"""Device-paced times of the prediction network's products on 1632 rows: 20 launches captured in ONE hipGraph and replayed
(tools/small_gemm_bench.py times host-paced calls: ~20 us of Python each, longer than these kernels).  Run in a tuning build
(PIKA_HIPCC_EXTRA=-DPIKA_TUNING_KNOBS python -m pika_amd.build --force) to compare tile / split-K overrides:
    python tools/small_gemm_graph_bench.py            (one subprocess per override)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("NN qkv 3 terms 1632x512x1536", 1632, 512, 1536, False), ("NN ffn1 1632x2048x1536", 1632, 2048, 1536, False),
          ("NN ffn2 1632x512x6144", 1632, 512, 6144, False), ("NN conv 1632x512x7680", 1632, 512, 7680, False),
          ("TN dW 512x512 r1632", 512, 512, 1632, True), ("TN dW 2048x512 r1632", 2048, 512, 1632, True)]


def worker():
    sys.path.insert(0, ROOT)
    import pika_amd  # noqa: F401
    import torch
    from pika_amd import gemm as G
    dev = torch.device("cuda:0")
    tag = os.environ.get("TAG", "")
    for name, M, N, K, trans in SHAPES:
        if trans:
            at, bt = torch.randn(K, M, device=dev).bfloat16(), torch.randn(K, N, device=dev).bfloat16()
            a_op, b_op = G.matrix_t(at)[0], G.matrix_t(bt)[0]
        else:
            a, b = torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16()
            a_op, b_op = G.matrix(a)[0], G.matrix(b)[0]
        out = torch.empty(M, N, device=dev)
        call = lambda: G.launch(a_op, b_op, out, N, M, N, K, precision="bf16")
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                call()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print("%-34s %-30s %7.1f us per launch" % (tag, name, e0.elapsed_time(e1) / 200 * 1e3), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        sp = {"PIKA_GEMM_SPLIT_MIN_NK": "8", "PIKA_GEMM_SPLIT_MIN_PER": "4"}
        runs = [{}, {"PIKA_GEMM_CFG": "1", "PIKA_GEMM_CFG_T": "1"}, dict(sp, PIKA_GEMM_SPLIT_TARGET="64"),
                dict(sp, PIKA_GEMM_SPLIT_TARGET="128"), dict(sp, PIKA_GEMM_SPLIT_TARGET="256"),
                dict(sp, PIKA_GEMM_SPLIT_TARGET="128", PIKA_GEMM_CFG="1", PIKA_GEMM_CFG_T="1")]
        for r in runs:
            env = dict(os.environ, TAG=",".join("%s=%s" % (k[10:], v) for k, v in r.items()) or "default", **r)
            subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=env, check=False)
