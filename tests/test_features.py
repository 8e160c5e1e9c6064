"""CMVN / SpecAugment: host RNG protocol vs golden bands recorded from the reference class
(tests/golden/make_specaug_golden.py), and the HIP kernels vs the reference formulas."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "specaug.npz")


def test_specaugment_draws_match_reference_rng_protocol():
    from pika_amd.features import SpecAugment
    z = np.load(GOLD)
    B, T, F = z["shape"]
    bands = z["bands"]
    i = 0
    while i < len(bands):
        s = int(bands[i][0])
        torch.manual_seed(s)
        np.random.seed(s)
        aug = SpecAugment(15, 35)
        for rep in range(2):
            _, _, f0, fs, t0, ts = [int(v) for v in bands[i]]
            got = aug.draw(int(T), int(F))
            # when a span is 0 the reference draws no start; our draw reports start 0 too
            assert got[1] == fs and got[3] == ts, (s, rep, got, bands[i])
            if fs > 0:
                assert got[0] == f0
            if ts > 0:
                assert got[2] == t0
            i += 1


@pytest.mark.gpu
def test_specaug_kernel_matches_reference_masks(hip_device):
    from pika_amd.features import SpecAugment
    z = np.load(GOLD)
    B, T, F = [int(v) for v in z["shape"]]
    for s in (0, 3, 11, 29):
        rows = z["bands"][z["bands"][:, 0] == s]
        torch.manual_seed(s)
        np.random.seed(s)
        aug = SpecAugment(15, 35)
        g = torch.Generator().manual_seed(s)
        for rep in range(2):
            base = torch.randn(B, T, F, generator=g) + 3.0
            x = base.to(hip_device)
            aug.apply(x)  # consumes the host RNG exactly like the reference
            _, _, f0, fs, t0, ts = [int(v) for v in rows[rep]]
            want = base.clone()
            want[:, :, f0:f0 + fs] = 0.0
            want[:, t0:t0 + ts, :] = 0.0
            assert torch.equal(x.cpu(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,F", [(1, 1, 1), (2, 7, 5), (3, 100, 240), (4, 1000, 240), (2, 33, 65)])
def test_cmvn_kernel(hip_device, B, T, F):
    from pika_amd.features import cmvn_apply_
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = torch.randn(B, T, F, generator=g) * 4 + 8
    off = torch.randn(F, generator=g)
    sc = torch.rand(F, generator=g) + 0.5
    # train_transducer_bmuf_otfaug.py:88-91 on the CPU, in float64 for the comparison
    ref = x.double()
    ref = ref - ref.mean(dim=1).unsqueeze(1)
    ref = (ref + off.double()) * sc.double()
    y = cmvn_apply_(x.to(hip_device).clone(), off.to(hip_device), sc.to(hip_device), cmn=True).cpu()
    assert torch.allclose(y.double(), ref, rtol=1e-5, atol=1e-5)
    # no CMN: exactly (x + offset) * scale in fp32
    y2 = cmvn_apply_(x.to(hip_device).clone(), off.to(hip_device), sc.to(hip_device), cmn=False).cpu()
    assert torch.equal(y2, (x + off) * sc)
    y3 = cmvn_apply_(x.to(hip_device).clone(), None, None, cmn=True).cpu()
    assert torch.allclose(y3.double(), x.double() - x.double().mean(dim=1, keepdim=True), atol=1e-5)
