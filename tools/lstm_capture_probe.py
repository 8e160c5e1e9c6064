"""Which form of the recipes' 2-layer LSTM prediction network survives hipGraph capture AND draws new inter-layer dropout
masks per replay (round 5, MI355X, torch 2.10 + MIOpen):   python tools/lstm_capture_probe.py fused|layered|layered_d|warm
  fused     nn.LSTM(num_layers=2, dropout=0.2): captures; every replay repeats the mask of the capture
  layered   one _VF.lstm call per layer with dropout 0 + torch dropout in between: the capture dies in hipBLASLt
  warm      the same after a warm-up on the capture stream: dies the same way
  layered_d one call per layer WITH the dropout argument (a no-op for one layer): captures, new masks per replay  <- shipped
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pika_amd
mode = sys.argv[1]
dev = torch.device("cuda:0")
torch.manual_seed(0)
rnn = torch.nn.LSTM(100, 1024, num_layers=2, dropout=0.2, batch_first=True).to(dev).train()
x = torch.randn(32, 51, 100, device=dev, requires_grad=True)
from pika_amd.model.transducer import _lstm_forward
def layered_d(rnn, x):
    out = x
    zeros = x.new_zeros(1, x.shape[0], rnn.hidden_size)
    for l in range(rnn.num_layers):
        w = [getattr(rnn, n % l) for n in ("weight_ih_l%d", "weight_hh_l%d", "bias_ih_l%d", "bias_hh_l%d")]
        out = torch._VF.lstm(out, (zeros, zeros), w, True, 1, rnn.dropout, True, False, True)[0]
        if l + 1 < rnn.num_layers:
            out = torch.nn.functional.dropout(out, rnn.dropout, True)
    return out
def fwd_bwd():
    if mode == "layered_d":
        y = layered_d(rnn, x)
    else:
        y = _lstm_forward(rnn, x) if mode != "fused" else rnn(x)[0]
    g = torch.autograd.grad(y.sum(), [x] + list(rnn.parameters()))
    return y
for _ in range(2):
    fwd_bwd()
torch.cuda.synchronize()
cs = torch.cuda.Stream()
if mode == "warm":
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        fwd_bwd()
    torch.cuda.current_stream().wait_stream(cs)
    torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=cs, capture_error_mode="thread_local"):
    y = fwd_bwd()
vals = []
for _ in range(3):
    g.replay()
    torch.cuda.synchronize()
    vals.append(float(y.sum()))
print(mode, "captured and replayed", vals)
