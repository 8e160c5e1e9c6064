"""Generates tests/golden/model_full_train.npz from the REFERENCE model code (trainer/model/transducer.py:73-112 imported
from /root/reference, CPU fp32): the FULL architecture of BASELINE.json configs[1] in TRAIN mode on the scenario of
tests/model_full_common.py.  Recorded: a strided slice of the encoder output and of the prediction-net output, a
strided sample of the log-probs, the RNN-T costs of the reference's log-probs (fp64 oracle, oracle/rnnt.py -- the
reference's own loss is the absent third-party warp_rnnt), and of EVERY parameter gradient of `costs.sum()` the
compact form of tests/golden/mbr_hooks.py (512 strided entries + L2 norm, sum, max, numel), plus the BatchNorm
running statistics after the step.  About a minute of CPU.
    python tests/golden/make_model_full_golden.py          # conv-transformer prediction net -> model_full_train.npz
    python tests/golden/make_model_full_golden.py rnn      # the recipes' 2-layer LSTM prediction net -> model_full_train_rnn.npz
    python tests/golden/make_model_full_golden.py long     # the BENCHMARKED length: B = 4, T_in = 1000 (T' = 240), U = 50 ->
                                                           # model_full_train_long.npz (same recorded quantities; ~2 min)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from oracle import pika_ref  # noqa: E402
from oracle import rnnt as O  # noqa: E402
import model_full_common as F  # noqa: E402
from mbr_hooks import compact  # noqa: E402

transducer = pika_ref.load_reference("trainer.model.transducer")
torch.set_num_threads(8)
ARG = sys.argv[1] if len(sys.argv) > 1 else "transformer"
assert ARG in ("transformer", "rnn", "long")
DEC = "rnn" if ARG == "rnn" else "transformer"
net = F.build(transducer, pika_ref.seeded_state_dict, DEC)
x, y, x_len, y_len = F.inputs(F.LONG if ARG == "long" else F.SHORT)
seen = {}
net.encoder.register_forward_hook(lambda m, i, o: seen.__setitem__("enc", o.detach()))
net.decoder.register_forward_hook(lambda m, i, o: seen.__setitem__("pred", (o[0] if isinstance(o, tuple) else o).detach()))
t0 = time.time()
lp = net(x, y, x_len, True)
costs, g = O.rnnt_loss(lp.detach().numpy(), y.numpy(), x_len.numpy(), y_len.numpy())      # fp64
lp.backward(torch.from_numpy(g.astype(np.float32)))
print("reference forward + backward: %.1f s; T' = %d; costs" % (time.time() - t0, lp.shape[1]), costs)
grads = {"g%03d" % i: p.grad.numpy() for i, (n, p) in enumerate(net.named_parameters())}
out = compact(grads)
out["n"] = np.array(len(grads))
out["names"] = np.array([n for n, _ in net.named_parameters()])
out["enc"] = F.enc_slice(seen["enc"]).numpy()
out["pred"] = seen["pred"][:, :, ::17].numpy()
out["lp"] = (F.lp_slice_long if ARG == "long" else F.lp_slice)(lp.detach()).numpy()
out["costs"] = costs
out["enc_absmax"] = np.array(float(seen["enc"].abs().max()))
for k in ("encoder.bn_in.running_mean", "encoder.hidden_bn.8.running_var", "encoder.bn_final.running_var"):
    out["buf:" + k] = net.state_dict()[k].numpy()
path = os.path.join(HERE, {"transformer": "model_full_train.npz", "rnn": "model_full_train_rnn.npz",
                          "long": "model_full_train_long.npz"}[ARG])
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KiB")
