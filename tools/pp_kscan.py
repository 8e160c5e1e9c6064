import sys, torch
sys.path.insert(0, "/root/repo")
from pika_amd import gemm as G
dev = torch.device("cuda:0")
def t(M, N, K, o16=False, n=5):
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if o16 else torch.float32)
    run = lambda: G.launch(G.matrix(a)[0], G.matrix(b)[0], out, N, M, N, K)
    run(); run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print("M %6d N %5d K %5d o16 %d: %7.3f ms %7.1f TF  per-tile-slot %6.2f us" % (M, N, K, o16, ms, 2.0*M*N*K/ms/1e9, ms*1e3/(tiles/256)), flush=True)
for K in (64, 256, 512, 1024, 2048, 4096):
    t(196608, 5120, K)
for K in (64, 256, 1024, 4096):
    t(196608, 5120, K, True)
for K in (64, 1024):
    t(196608, 1024, K)
    t(32768, 5120, K)
x = torch.empty(196608, 5120, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
x.fill_(1.0); e0.record()
for _ in range(5): x.fill_(2.0)
e1.record(); torch.cuda.synchronize(); print("fill 4 GB: %.3f ms" % (e0.elapsed_time(e1)/5))
