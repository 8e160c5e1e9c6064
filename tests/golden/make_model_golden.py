"""Generates tests/golden/model_tiny_{transformer,rnn}.npz from the REFERENCE model code
(imported from /root/reference, CPU, fp32): encoder / prediction-net / joint activations in
eval and train mode and parameter gradients of a fixed weighted-sum probe.
    python tests/golden/make_model_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pika_ref  # noqa: E402
import model_common as C  # noqa: E402

transducer, encoder = pika_ref.load_reference("trainer.model.transducer",
                                              "trainer.model.rnnt_tdnn_transformer")
GRAD_KEYS = ["encoder.fc_in.weight", "encoder.hidden_conv.0.weight", "encoder.hidden_conv.5.bias",
             "encoder.hidden_bn.2.weight", "encoder.transformer.0.self_attn.linear_keys.weight",
             "encoder.transformer.1.feed_forward.w_2.bias", "encoder.transformer.0.layer_norm.weight",
             "encoder.fc_out.weight", "embed.weight", "fc1.weight", "fc_gate.bias", "fc2.weight"]

for dec in ("transformer", "rnn"):
    net = C.build(transducer, encoder, dec)
    net.load_state_dict(pika_ref.seeded_state_dict(net, C.SEED))
    x, y, y_len, w = C.inputs()
    out = {}
    net.eval()
    with torch.no_grad():
        out["enc_eval"] = net.encoder(x).numpy()
        sos = torch.zeros(C.B, 1, dtype=torch.long)
        ys = torch.cat((sos, y), 1)
        out["pred_eval"] = (net.decoder(net.embed(ys))[0] if dec == "rnn" else net.decoder(ys)).numpy()
        out["joint_eval"] = net(x, y, None, True).numpy()
        out["joint_eval_nosm"] = net(x, y, None, False).numpy()
    net.train()
    lp = net(x, y, None, True)
    out["joint_train"] = lp.detach().numpy()
    (lp * w).sum().backward()
    params = dict(net.named_parameters())
    keys = [k for k in GRAD_KEYS if k in params] + \
           (["decoder.conv.0.weight", "decoder.transformer.1.self_attn.final_linear.bias",
             "decoder.linear_out.weight"] if dec == "transformer" else ["decoder.weight_hh_l1"])
    for k in keys:
        out["grad:" + k] = params[k].grad.numpy()
    out["grad_keys"] = np.array(keys)
    out["bn_in_running_mean_after"] = net.encoder.bn_in.running_mean.numpy()
    out["bn_final_running_var_after"] = net.encoder.bn_final.running_var.numpy()
    path = os.path.join(HERE, "model_tiny_%s.npz" % dec)
    np.savez_compressed(path, **out)
    print(dec, "T' =", out["enc_eval"].shape[1], "->", path, os.path.getsize(path) // 1024, "KiB")
