"""pika_dfc2_topk (the search step's vocabulary product + partial log-sum-exp / top-K) alone: time per launch at the
search's row counts, one subprocess per PIKA_DFC2_BM.    GPU box: python tools/dfc2_bench.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    sys.path.insert(0, ROOT)
    import pika_amd  # noqa: F401
    import torch
    from pika_amd import _lib
    from pika_amd.decoder.fused_step import PackedWeight
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    V, Hd, K, sm = 5000, 1024, 16, 0.8
    W = (torch.randn(V, Hd) * 0.05).to(dev)
    bias = torch.randn(V).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    for terms in (4, 2):
        pw = PackedWeight(W, terms)
        for R in (1024, 512, 128, 32):
            h = torch.randn(R, Hd).to(dev)
            splits = lib.pika_dfc2_splits(V)
            pmax = torch.empty(R * splits, device=dev)
            psum = torch.empty(R * splits, device=dev)
            pcand = torch.empty(R * splits * K * 8, dtype=torch.uint8, device=dev)

            def fn():
                _lib.check(lib.pika_dfc2_topk(h.data_ptr(), Hd, pw.buf.data_ptr(), bias.data_ptr(), R, V, Hd, terms, sm, K,
                                              pmax.data_ptr(), psum.data_ptr(), pcand.data_ptr(), st), "pika_dfc2_topk")
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print("BM %-4s terms %d rows %4d: %7.1f us   (checksum %.6e)" % (
                os.environ.get("PIKA_DFC2_BM", "auto"), terms, R, e0.elapsed_time(e1) / 50 * 1e3,
                float(pmax.double().sum() + psum.double().sum())), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        for bm in ("32", "64"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=dict(os.environ, PIKA_DFC2_BM=bm),
                           check=False, timeout=200)
