"""Times the fused-epilogue products of the transformer layers (ffn1 with ReLU + dropout -> bf16, ffn2 / attention
projection with dropout + residual, the masked dX product) on the GPU box:  python tools/pp_epi_bench.py"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pika_amd.model import hipops as H   # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


M = 31232
for name, N, K, kind in (("ffn1  relu+dropout->bf16", 4096, 1024, "epi1"), ("ffn2  dropout+residual", 1024, 4096, "epi3"),
                         ("proj  dropout+residual", 1024, 1024, "epi3"), ("ffn dX  mask->bf16", 1024, 4096, "epi2"),
                         ("ffn1  plain bf16 (p=0)", 4096, 1024, "epi1p0")):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    if kind in ("epi1", "epi1p0"):
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        p = 0.1 if kind == "epi1" else 0.0
        fn = lambda: H._gemm_epilogue(a, b, out, bias, 1, relu=1, p_drop=p, seed=7)
    elif kind == "epi2":
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        aux = torch.randn(M, N, device=dev).bfloat16()
        fn = lambda: H._gemm_epilogue(a, b, out, None, 2, aux=aux, scale=1.1)
    else:
        out = torch.empty(M, N, device=dev)
        res = torch.randn(M, N, device=dev)
        fn = lambda: H._gemm_dropout_residual(a, b, out, bias, 0.1, 7, res)
    ms = timeit(fn)
    print("%-28s M %d N %d K %d: %7.3f ms  %7.1f TFLOP/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
