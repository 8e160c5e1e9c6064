"""pika_dfc2_logits (the search step's vocabulary product + row statistics) alone: time per launch at the search's row
counts with a 64 MB copy between launches (as inside a step, W does not survive in L2), one subprocess per knob value of a
TUNING build (PIKA_HIPCC_EXTRA=-DPIKA_TUNING_KNOBS python -m pika_amd.build --force).
    GPU box: python tools/dfc2_bench.py [PIKA_DFC2_MAP rows cols | PIKA_DFC2_BM 32 64]"""
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
            ldl = splits * lib.pika_dfc2_cols_per_split()
            logits = torch.empty(R, ldl, device=dev)
            evict_a, evict_b = torch.empty(8 << 20, device=dev), torch.empty(8 << 20, device=dev)

            def fn():
                _lib.check(lib.pika_dfc2_logits(h.data_ptr(), Hd, pw.buf.data_ptr(), bias.data_ptr(), R, V, Hd, terms, sm,
                                                pmax.data_ptr(), psum.data_ptr(), logits.data_ptr(), ldl, st), "pika_dfc2_logits")
            for _ in range(5):
                fn()
            tot = 0.0
            for _ in range(30):
                evict_b.copy_(evict_a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            us = tot / 30 * 1e3
            print("%s terms %d rows %4d: %7.1f us (one launch between two events, incl. ~4 us of event overhead)   (checksum %.6e)" % (
                os.environ.get("DFC2_TAG", ""), terms, R, us, float(pmax.double().sum() + psum.double().sum())), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        var = sys.argv[1] if len(sys.argv) > 1 else "PIKA_DFC2_MAP"
        for val in (sys.argv[2:] or ["rows", "cols"]):
            subprocess.run([sys.executable, os.path.abspath(__file__), "worker"],
                           env=dict(os.environ, **{var: val, "DFC2_TAG": "%s=%s" % (var, val)}), check=False, timeout=300)
