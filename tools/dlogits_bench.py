"""d(logits) kernel (pika_rnnt_dlogits_compact_bf16 with column sums) at the config-2 lattice under PIKA_DLOGITS_RPW."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    import torch
    sys.path.insert(0, ROOT)
    from pika_amd import _lib, rnnt as R
    dev = torch.device("cuda:0")
    B, T, U1, V = 32, 240, 51, 5000
    logits = torch.randn(B, T, U1, V, device=dev)
    labels = torch.randint(1, V, (B, U1 - 1), device=dev, dtype=torch.int32)
    tl = torch.full((B,), T, dtype=torch.int32, device=dev); ul = torch.full((B,), U1 - 1, dtype=torch.int32, device=dev)
    lib = _lib.lib()
    ws = torch.empty(lib.pika_rnnt_workspace_bytes(B, T, U1), dtype=torch.uint8, device=dev)
    costs = torch.empty(B, device=dev); lse = torch.empty(B * T * U1, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.pika_rnnt_fused_forward(logits.data_ptr(), labels.data_ptr(), tl.data_ptr(), ul.data_ptr(), B, T, U1, V, 0,
                                           costs.data_ptr(), lse.data_ptr(), ws.data_ptr(), st), "fwd")
    gc = torch.ones(B, device=dev)
    _lib.check(lib.pika_rnnt_loss_backward(labels.data_ptr(), tl.data_ptr(), ul.data_ptr(), B, T, U1, V, 0, gc.data_ptr(),
                                           ws.data_ptr(), None, st), "bwd")
    out = torch.empty(B * T * U1, 5056, dtype=torch.bfloat16, device=dev); db = torch.empty(V, device=dev)
    def run():
        _lib.check(lib.pika_rnnt_dlogits_compact_bf16(logits.data_ptr(), lse.data_ptr(), ws.data_ptr(), B, T, U1, V, 0,
                                                      out.data_ptr(), 5056, 1.0, db.data_ptr(), st), "dlogits")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("rpw=%s narrow=%s  %.3f ms  %.2f TB/s" % (os.environ.get("PIKA_DLOGITS_RPW"), os.environ.get("PIKA_DLOGITS_NARROW"), ms,
                                                    (logits.numel() * 4 + out.numel() * 2) / ms / 1e9), flush=True)
else:
    for env in ({"PIKA_DLOGITS_RPW": "16"}, {"PIKA_DLOGITS_RPW": "32"}, {"PIKA_DLOGITS_RPW": "64"}, {"PIKA_DLOGITS_RPW": "128"},
                {"PIKA_DLOGITS_RPW": "256"}, {"PIKA_DLOGITS_RPW": "32", "PIKA_DLOGITS_NARROW": "1"}):
        subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=dict(os.environ, **env))
