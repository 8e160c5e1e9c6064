"""What the vendor library (hipBLASLt through torch.mm, bf16 operands, fp32 output) takes on the prediction network's products
on 1632 rows, next to the register-staged kernel's times (tools/small_gemm_bench.py).  GPU box: python tools/small_gemm_vendor_probe.py"""
import torch

dev = torch.device("cuda:0")
shapes = [("qkv/out 1632x512x1536(3 terms)", 1632, 512, 1536), ("ffn1 1632x2048x1536", 1632, 2048, 1536),
          ("ffn2 1632x512x6144", 1632, 512, 6144), ("conv 1632x512x7680", 1632, 512, 7680),
          ("one term 1632x512x512", 1632, 512, 512), ("one term 1632x2048x512", 1632, 2048, 512)]


def timeit(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3


for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev)
    try:
        t32 = timeit(lambda: torch.mm(a, b.t(), out_dtype=torch.float32))
    except Exception as e:
        t32 = float("nan")
        print("out_dtype=float32:", type(e).__name__, str(e)[:100])
    t16 = timeit(lambda: torch.mm(a, b.t()))
    print("%-34s fp32 out %7.1f us   bf16 out %7.1f us" % (name, t32, t16), flush=True)
