"""Times the fused attention kernels against the unfused torch chain at the encoder's shapes."""
import sys
import time

import torch

sys.path.insert(0, ".")
from pika_amd import gemm as G  # noqa: E402
from pika_amd.model import ops  # noqa: E402


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    for B, T, H, D in [(32, 988, 16, 64), (32, 976, 16, 64), (32, 958, 8, 128)]:
        q, k, v = [torch.randn(B, T, H * D, device=dev, requires_grad=True) for _ in range(3)]
        w = torch.randn(B, T, H * D, device=dev)
        flops = 4.0 * B * H * T * T * D
        for mode in ("bf16", "fp32"):
            G.PRECISION = mode

            def fwd():
                return ops.attention(q, k, v, H, None, 0.2, True)

            def fwdbwd():
                out = ops.attention(q, k, v, H, None, 0.2, True)
                out.backward(w)
                q.grad = k.grad = v.grad = None
            tf, tb = timeit(fwd), timeit(fwdbwd)
            print("B=%d T=%d H=%d D=%d  %s: fwd %.3f ms (%.0f TFLOP/s)  fwd+bwd %.3f ms (%.0f TFLOP/s)" % (
                B, T, H, D, "fused" if mode == "bf16" else "torch", tf, flops / tf / 1e9, tb, 3.5 * flops / tb / 1e9))


if __name__ == "__main__":
    main()
