"""LayerNorm backward at the encoder's shape under PIKA_LN_BWD_RPB (rows per workgroup = column-sum atomics per row)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    import torch
    sys.path.insert(0, ROOT)
    from pika_amd.model.hipops import LayerNormFn
    dev = torch.device("cuda:0")
    x = torch.randn(32, 994, 1024, device=dev, requires_grad=True)
    w = torch.ones(1024, device=dev, requires_grad=True); b = torch.zeros(1024, device=dev, requires_grad=True)
    y = LayerNormFn.apply(x, w, b, 1e-6, True)
    g = torch.randn_like(y)
    for _ in range(3):
        y.backward(g, retain_graph=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y.backward(g, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    print("rpb=%s  %.1f us per backward (incl. 2 memsets + grad accumulation adds)" % (os.environ.get("PIKA_LN_BWD_RPB"), e0.elapsed_time(e1) / 20 * 1e3), flush=True)
else:
    for r in ("8", "16", "32", "64", "128", "256"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=dict(os.environ, PIKA_LN_BWD_RPB=r))
