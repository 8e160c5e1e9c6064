"""The joint's logits product with its 16-bit epilogue (gemm_pp<4>: pika_gemm_bf16_nt_lse_f16) at the benchmark shape
(391680 x 5000 x 1024), next to the same product with a plain bf16 epilogue and the same FLOPs in the dh orientation:
    python tools/fc2_epi_bench.py"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pika_amd import _lib, gemm as G   # noqa: E402

dev = torch.device("cuda:0")
B, T, U1, V, K = 32, 240, 51, 5000, 1024
M = B * T * U1


def timeit(fn, n=5):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


h = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
w = (torch.randn(V, K, device=dev) * 0.05).bfloat16()
bias = torch.randn(V, device=dev)
labels = torch.randint(1, V, (B, U1 - 1), device=dev, dtype=torch.int32)
out16 = torch.empty(M, V, dtype=torch.float16, device=dev)
n_part = (V + 255) // 256 * 4
part = torch.empty(2, M, n_part, device=dev)
gath = torch.empty(M, 2, device=dev)
lib = _lib.lib()
st = torch.cuda.current_stream().cuda_stream


def fc2():
    _lib.check(lib.pika_gemm_bf16_nt_lse_f16(h.data_ptr(), K, w.data_ptr(), K, out16.data_ptr(), V, M, V, K, bias.data_ptr(),
                                             part[0].data_ptr(), part[1].data_ptr(), n_part, labels.data_ptr(), T, U1, 0,
                                             gath.data_ptr(), st), "lse_f16")


ms = timeit(fc2)
print("fc2 16-bit epilogue (EPI 4)     %7.3f ms  %7.1f TFLOP/s" % (ms, 2.0 * M * V * K / ms / 1e9), flush=True)
outb = torch.empty(M, V, dtype=torch.bfloat16, device=dev)
ms = timeit(lambda: G.launch(G.matrix(h)[0], G.matrix(w)[0], outb, V, M, V, K, bias=bias))
print("same product, plain bf16 out    %7.3f ms  %7.1f TFLOP/s" % (ms, 2.0 * M * V * K / ms / 1e9), flush=True)
del outb, out16, part
dy = torch.randn(M, 5056, device=dev).bfloat16()
wt = (torch.randn(K, 5056, device=dev) * 0.05).bfloat16()
dh = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
ms = timeit(lambda: G.launch(G.matrix(dy)[0], G.matrix(wt)[0], dh, K, M, K, 5056))
print("dh orientation (1024 outputs)   %7.3f ms  %7.1f TFLOP/s" % (ms, 2.0 * M * 5056 * K / ms / 1e9), flush=True)
