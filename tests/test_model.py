"""Model parity: our encoder / prediction net / joint against golden
activations and gradients recorded from the REFERENCE model code (tests/golden/
make_model_golden.py, PyTorch-CPU fp32).  Identical weights on both sides come from
oracle.pika_ref.seeded_state_dict (same keys and shapes => same tensors).
Tolerance: 1e-3 rel fp32 per north_star; measured error is ~1e-6 (fp32 reassociation only)."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import model_common as C  # noqa: E402
from oracle.pika_ref import seeded_state_dict  # noqa: E402  (deterministic weights only)


def ours(dec, device="cpu"):
    from pika_amd.model import transducer, encoder
    net = C.build(transducer, encoder, dec)
    net.load_state_dict(seeded_state_dict(net, C.SEED))
    return net.to(device)


def close(a, b, rtol=1e-3, atol=1e-5):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    scale = np.abs(b).max()
    err = np.abs(a - b).max()
    assert err <= rtol * scale + atol, "max err %g vs scale %g" % (err, scale)


def close_in_norm(a, b, rtol, floor):
    """||a - b|| <= rtol * max(||b||, floor): the measure for gradients of a ReLU network under a different rounding of
    the forward pass (a pre-activation within the forward error of zero flips its mask and changes ONE gradient path
    outright: large in the max norm of a small tensor, small in the L2 norm)."""
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    err, ref = np.linalg.norm(a - b), max(np.linalg.norm(b), floor)
    assert err <= rtol * ref, "||err|| %g vs ||ref|| %g" % (err, ref)


def run_parity(dec, device, act_rtol=1e-3, grad_norm_rtol=None, joint_rtol=None):
    z = np.load(os.path.join(HERE, "golden", "model_tiny_%s.npz" % dec))
    net = ours(dec, device)
    x, y, y_len, w = [t.to(device) for t in C.inputs()]
    net.eval()
    with torch.no_grad():
        close(net.encoder(x), z["enc_eval"], rtol=act_rtol)
        sos = torch.zeros(C.B, 1, dtype=torch.long, device=device)
        close(net.predict(torch.cat((sos, y), 1)), z["pred_eval"], rtol=act_rtol)
        joint_rtol = act_rtol if joint_rtol is None else joint_rtol
        close(net(x, y, None, True), z["joint_eval"], rtol=joint_rtol)
        close(net(x, y, None, False), z["joint_eval_nosm"], rtol=joint_rtol)
    net.train()
    lp = net(x, y, None, True)
    close(lp, z["joint_train"], rtol=joint_rtol)
    (lp * w).sum().backward()
    params = dict(net.named_parameters())
    gscale = max(float(np.linalg.norm(z["grad:" + str(k)])) for k in z["grad_keys"])
    for k in z["grad_keys"]:
        if grad_norm_rtol is None:
            close(params[str(k)].grad, z["grad:" + str(k)], rtol=1e-3, atol=1e-5)
        else:
            close_in_norm(params[str(k)].grad, z["grad:" + str(k)], grad_norm_rtol, 1e-3 * gscale)
    close(net.encoder.bn_in.running_mean, z["bn_in_running_mean_after"])
    close(net.encoder.bn_final.running_var, z["bn_final_running_var_after"])


@pytest.mark.parametrize("dec", ["transformer", "rnn"])
def test_module_tree_matches_reference_layout(dec):
    """state_dict keys/shapes are the checkpoint + BMUF-vector contract."""
    net = ours(dec)
    z = np.load(os.path.join(HERE, "golden", "model_tiny_%s.npz" % dec))
    names = [k for k, _ in net.named_parameters()]
    for k in z["grad_keys"]:
        assert str(k) in names
    for attr in ("encoder", "decoder", "embed", "fc1", "fc_gate", "fc2", "pack_seq", "decoder_type",
                 "hid_dim", "output_dim"):
        assert hasattr(net, attr)
    assert net.embed.padding_idx == C.V and net.fc1.in_features == 2 * C.H


@pytest.mark.parametrize("dec", ["transformer", "rnn"])
def test_cpu_plumbing_matches_reference_golden(dec):
    run_parity(dec, "cpu")


def test_full_size_parameter_count_and_pickle_roundtrip(tmp_path):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    import importlib
    mod = importlib.import_module("model.transducer")  # how the training script finds it
    opt = C.make_opt("transformer")
    opt.rnn_size, opt.embd_dim, opt.padding_idx = 1024, 100, 5000
    net = mod.Net(opt, 240, 5000)
    assert sum(p.numel() for p in net.parameters()) == 85648652  # SURVEY 2.3 [probe]: 85.6 M
    small = ours("transformer")
    f = tmp_path / "model.epoch.0.0"
    torch.save(small, f)  # whole-module pickle, as train_transducer_bmuf_otfaug.py:363-366
    back = torch.load(f, weights_only=False)
    assert all(torch.equal(a, b) for a, b in zip(small.state_dict().values(), back.state_dict().values()))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "mixed"])
@pytest.mark.parametrize("dec", ["transformer", "rnn"])
def test_gpu_matches_reference_golden(hip_device, dec, mode):
    """The two arithmetic modes that carry the north_star tolerance (activations and loss within 1e-3 rel of the
    reference's fp32): "fp32" (exact three-term split, 6 MFMAs: activations AND every recorded gradient at 1e-3, measured
    1e-6) and "bf16x3" (two terms per operand, one bf16 product over a three times longer reduction on the
    direct-to-LDS kernels: activations held to 1e-4 here, measured 1.4e-5).  Gradients in the second mode: products are
    accurate to 1e-5, but a forward pass that differs by 1e-5 flips the ReLU mask of the few pre-activations that close
    to zero, and each flip changes one gradient path outright -- 1e-2 in the max norm of a (64, 240) tensor from ONE
    element; tools/precision_table.py shows the exact mode doing the same under a 1e-5 input perturbation.  So they are
    held in the L2 norm."""
    from pika_amd import gemm as G
    old, old_joint = G.PRECISION, G.X3_JOINT_BF16
    G.PRECISION = mode
    G.X3_JOINT_BF16 = False          # every product in two terms, the joint's lattice products included
    try:
        n0 = G.BF16X3_STATS["fast"]
        if mode == "fp32":
            run_parity(dec, hip_device)
        elif mode == "mixed":
            # the train-step default: two-term forward (activations as in "bf16x3"), bf16 backward (gradients in the L2 norm
            # at the bf16 operand budget)
            # at the bf16 operand budget); the joint's lattice product (fc2) runs on ONE bf16 term in this mode: logits to 4e-3 of their scale
            run_parity(dec, hip_device, act_rtol=1e-4, grad_norm_rtol=6e-2, joint_rtol=4e-3)
        else:
            run_parity(dec, hip_device, act_rtol=1e-4, grad_norm_rtol=3e-2)
            assert G.BF16X3_STATS["fast"] > n0 + 20
    finally:
        G.PRECISION, G.X3_JOINT_BF16 = old, old_joint


@pytest.mark.gpu
def test_gpu_bf16_arithmetic_stays_close(hip_device):
    """Config-2 arithmetic (bf16 operands, fp32 accumulate): activations within 3e-2 of the fp32
    reference relative to the tensor scale (documented tolerance of the bf16 mode)."""
    from pika_amd import gemm as G
    old = G.PRECISION
    G.PRECISION = "bf16"
    try:
        z = np.load(os.path.join(HERE, "golden", "model_tiny_transformer.npz"))
        net = ours("transformer", hip_device)
        x, y, y_len, w = [t.to(hip_device) for t in C.inputs()]
        net.eval()
        with torch.no_grad():
            close(net.encoder(x), z["enc_eval"], rtol=3e-2)
            close(net(x, y, None, True), z["joint_eval"], rtol=3e-2)
    finally:
        G.PRECISION = old
