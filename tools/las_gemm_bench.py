"""The four products of a LAS scoring token (pika_dgemm on a gather list of m rows out of n_max, two fp16 terms per
operand) and its attention launch, timed back to back on one MI355X at the row counts a pass goes through.
    python tools/las_gemm_bench.py            # one subprocess per PIKA_DGEMM_WIDE setting
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, E, NMAX, S, B = 1024, 100, 960, 240, 64
SHAPES = [("gates L0 4096x%d" % (E + 2 * H), 4 * H, E + 2 * H), ("gates L1 4096x2048", 4 * H, 2 * H),
          ("query    1024x1024", H, H), ("out      1024x2048", H, 2 * H)]
ROWS = [64, 256, 470, 700, 960]


def worker():
    import torch
    sys.path.insert(0, ROOT)
    from pika_amd import _lib
    from pika_amd.decoder.fused_step import DGemm, PackedWeight
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    tag = os.environ.get("TAG", "")
    g = torch.Generator().manual_seed(1)
    st = torch.cuda.current_stream().cuda_stream

    def timeit(fn, n=30):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    md = torch.zeros(1, dtype=torch.int32, device=dev)
    off = torch.zeros(1, dtype=torch.int32, device=dev)
    rl = torch.randperm(NMAX, generator=g).to(torch.int32).to(dev)
    for name, N, K in SHAPES:
        Kp = (K + 31) // 32 * 32
        A = torch.randn(NMAX, Kp, generator=g).to(dev)
        A[:, K:] = 0
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        C = torch.zeros(NMAX, N, device=dev)
        pw = PackedWeight(W, 4)
        d = DGemm()
        d.A, d.lda, d.W, d.bias, d.C, d.ldc = A.data_ptr(), Kp, pw.buf.data_ptr(), bias.data_ptr(), C.data_ptr(), N
        d.M, d.N, d.K, d.terms, d.flags = NMAX, N, K, 4, 0
        d.m_dev, d.rowlist, d.rowoff_dev = md.data_ptr(), rl.data_ptr(), off.data_ptr()
        d.skip_node = -1
        line = []
        for m in ROWS:
            md.fill_(m)
            if os.environ.get("BENCH_EXACT") == "1":        # launch sized for m rows (no idle workgroups)
                d.M, d.m_dev = m, None
            if os.environ.get("BENCH_NOLIST") == "1":       # rows [0, m) instead of a gather list
                d.rowlist, d.rowoff_dev = None, None
            us = timeit(lambda: _lib.check(lib.pika_dgemm(ctypes.byref(d), st), "pika_dgemm"))
            line.append("%4d: %6.1f" % (m, us))
        rows = rl[:ROWS[-1]].long() if d.rowlist else torch.arange(ROWS[-1], device=dev)
        want = A[rows][:, :K].double() @ W.double().t() + bias.double()
        err = ((C[rows].double() - want).abs().max() / want.abs().max()).item()
        print("%-36s %-22s %s us   (max err %.1e)" % (tag, name, "  ".join(line), err), flush=True)
    # the attention launch of a token: m queries of B utterances (list ordered by utterance), S positions
    wq = torch.randn(NMAX, H, generator=g).to(dev)
    proj = torch.randn(B, S, H, generator=g).to(dev)
    ctx = torch.randn(B, S, H, generator=g).to(dev)
    own = (torch.arange(NMAX) % B).sort().values.to(torch.int32).to(dev)
    lens = torch.full((B,), S, dtype=torch.int32, device=dev)
    v = torch.randn(H, generator=g).to(dev)
    out = torch.zeros(NMAX, 2 * H, device=dev)
    line = []
    for m in ROWS:
        md.fill_(m)
        # m rows spread over all utterances, ordered by utterance
        pick = torch.sort(torch.randperm(NMAX, generator=g)[:m]).values.to(torch.int32).to(dev)
        us = timeit(lambda: _lib.check(lib.pika_las_mlp_attention(
            wq.data_ptr(), H, proj.data_ptr(), ctx.data_ptr(), own.data_ptr(), lens.data_ptr(), pick.data_ptr(),
            v.data_ptr(), out.data_ptr(), 2 * H, None, NMAX, B, S, H, md.data_ptr(), off.data_ptr(), st), "attention"))
        line.append("%4d: %6.1f" % (m, us))
    print("%-36s %-22s %s us" % (tag, "mlp attention S=%d" % S, "  ".join(line)), flush=True)
    # ... utterance by utterance (chunk launch + merge launch)
    import numpy as np
    work = torch.empty(int(lib.pika_las_attention_work_floats(NMAX, S, H)), device=dev)
    st_d = torch.zeros(1, dtype=torch.int32, device=dev)
    line, errs = [], []
    for m in ROWS:
        md.fill_(m)
        pick = torch.sort(torch.randperm(NMAX, generator=g)[:m]).values.to(torch.int32)
        uoff = torch.from_numpy(np.searchsorted(own.cpu().numpy()[pick.numpy()], np.arange(B + 1)).astype(np.int32)).to(dev)
        pick = pick.to(dev)
        ref = torch.zeros(NMAX, 2 * H, device=dev)
        _lib.check(lib.pika_las_mlp_attention(
            wq.data_ptr(), H, proj.data_ptr(), ctx.data_ptr(), own.data_ptr(), lens.data_ptr(), pick.data_ptr(),
            v.data_ptr(), ref.data_ptr(), 2 * H, None, NMAX, B, S, H, md.data_ptr(), off.data_ptr(), st), "attention")
        out.zero_()
        us = timeit(lambda: _lib.check(lib.pika_las_mlp_attention_by_utterance(
            wq.data_ptr(), H, proj.data_ptr(), ctx.data_ptr(), own.data_ptr(), lens.data_ptr(), pick.data_ptr(),
            uoff.data_ptr(), v.data_ptr(), out.data_ptr(), 2 * H, work.data_ptr(), NMAX, B, S, H, md.data_ptr(),
            off.data_ptr(), st_d.data_ptr(), st), "attention by utterance"))
        line.append("%4d: %6.1f" % (m, us))
        errs.append((out - ref).abs().max().item())
    print("%-36s %-22s %s us   (max diff to the per-query kernel %.1e)" % (tag, "... by utterance", "  ".join(line), max(errs)),
          flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        runs = [{"PIKA_DGEMM_WIDE": "0"}, {"PIKA_DGEMM_WIDE": "1"}]
        for r in runs:
            env = dict(os.environ, TAG=",".join("%s=%s" % (k.replace("PIKA_DGEMM_", ""), v) for k, v in r.items()), **r)
            subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=env, check=False)
