"""Reference point for the hand-written GEMMs: the vendor library (torch.matmul = hipBLASLt) on the same bf16 shapes as
tools/pp_bench.py, same box, same clocks.  Not used by the product.    python tools/blaslt_ref.py"""
import torch

SHAPES = [("joint_logits", 391680, 5000, 1024), ("joint_dh", 391680, 1024, 5056), ("ffn1", 31616, 4096, 1024),
          ("ffn2", 31616, 1024, 4096), ("proj", 31616, 1024, 1024), ("tdnn", 31616, 1024, 3072), ("square", 8192, 8192, 8192)]
dev = torch.device("cuda:0")
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    for _ in range(2):
        c = a @ b.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5 if M > 100000 else 20
    e0.record()
    for _ in range(n):
        c = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("hipBLASLt bf16 (bf16 out) %-13s %8.3f ms  %7.1f TFLOP/s" % (name, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    del a, b, c
