"""Full-WIDTH LAS rescoring parity (SURVEY 8a row 16 at the width bench.py's configs[4] leg runs): per-token log-probs of
the REFERENCE `las.Net` scored one hypothesis at a time (tests/golden/make_las_full_golden.py) against ONE batched
`score_nbest_batch` pass per rescorer -- on the GPU in the PACKAGE-DEFAULT arithmetic, i.e. with everything the
benchmarked pass uses on the path: the persistent BLSTM kernel (pika_blstm_layer, two bf16 terms of W_hh in registers),
the captured token loop on two fp16 terms per operand, prefix sharing, the padded ragged batch.

Tolerance on a log-prob (values span -19 .. -1e-3): 1e-4 absolute.  The budget: the BLSTM recurrence keeps W_hh as two
bf16 terms (2^-17 per product, 61 dependent steps, two layers: 2e-5 on the encoder outputs, tests/test_las_kernels_gpu.py),
the token loop's products are fp32-grade (2^-22), and dec_proj is sharpened 30x in this scenario so that errors of the
decoder state SHOW.  Measured on MI355X: 5.7e-6 (batched pass, fw and bw), 7.6e-6 (one hypothesis at a time)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import las_full_common as LF  # noqa: E402
from oracle.pika_ref import seeded_state_dict  # noqa: E402

GOLD = os.path.join(HERE, "golden", "las_full.npz")


def nets(device):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pika_amd", "dropin"))
    from trainer.model import las
    return (LF.build(las, seeded_state_dict, LF.SEED_FW).to(device), LF.build(las, seeded_state_dict, LF.SEED_BW).to(device))


def worst(got, z, key, hyps):
    w = 0.0
    for b, row in enumerate(hyps):
        assert len(got[b]) == len(row)
        for j in range(len(row)):
            want = z["%s/%d/%d" % (key, b, j)]
            assert len(got[b][j]) == len(want) == len(row[j]) + 1
            w = max(w, float(np.abs(np.asarray(got[b][j]) - want).max()))
    return w


def test_cpu_full_width_las_scores_match_reference():
    z = np.load(GOLD)
    fw, bw = nets("cpu")
    src, lens, hyps = LF.inputs()
    assert worst(fw.score_nbest_batch(src, lens, hyps, LF.SOS, LF.EOS), z, "fw", hyps) < 2e-4
    rev = [[h[::-1] for h in row] for row in hyps]
    assert worst(bw.score_nbest_batch(src, lens, rev, LF.SOS, LF.EOS), z, "bw", hyps) < 2e-4


@pytest.mark.gpu
def test_gpu_full_width_las_scores_match_reference_in_the_benchmarked_arithmetic(hip_device):
    from pika_amd import gemm as G
    assert G.PRECISION == "mixed", "this test pins the package default (what bench.py's decode leg rescoring runs in)"
    z = np.load(GOLD)
    fw, bw = nets(hip_device)
    src, lens, hyps = LF.inputs()
    src = src.to(hip_device)
    rev = [[h[::-1] for h in row] for row in hyps]
    out = {}
    for key, net, lists in (("fw", fw, hyps), ("bw", bw, rev)):
        net.encoder._status = None
        got = net.score_nbest_batch(src, lens, lists, LF.SOS, LF.EOS)
        # everything the benchmarked pass runs was on this path:
        assert net.encoder._status is not None, "the persistent BLSTM kernel did not take the encoder pass"
        assert net.last_pass["shared"] and net.last_pass["row_steps"] < net.last_pass["pairs"], net.last_pass
        assert getattr(net.decoder, "fused_terms", 4) == 4 and os.environ.get("PIKA_LAS_GRAPH", "1") != "0"
        out[key] = worst(got, z, key, hyps)
    print("full-width LAS rescoring, default arithmetic: max |log-prob - reference| fw %.2e, bw %.2e "
          "(row steps %d of %d pairs)" % (out["fw"], out["bw"], fw.last_pass["row_steps"], fw.last_pass["pairs"]))
    assert out["fw"] < 1e-4 and out["bw"] < 1e-4, out
    # the unchanged decode script asks one hypothesis at a time (decode_transducer.py:136-156): the same values through
    # TransducerDecoder.las_rescore on the B = 1 encoder pass (persistent kernel at B = 1, no sharing)
    from types import SimpleNamespace
    from decoder.transducer_decoder import TransducerDecoder
    d = TransducerDecoder(None, 1, 1, args=SimpleNamespace(las_rescorer=fw, las_rescorer_bw=bw, bilas_rescorer=None))
    w1 = 0.0
    for b in (0, 3):
        x = src[:lens[b], b:b + 1].contiguous()
        for j in (0, 3, 7):
            tgt = torch.LongTensor([LF.SOS] + hyps[b][j] + [LF.EOS]).to(hip_device).unsqueeze(-1).unsqueeze(-1)
            w1 = max(w1, float(np.abs(np.asarray(d.las_rescore(x, tgt)) - z["fw/%d/%d" % (b, j)]).max()))
            tgt = torch.LongTensor([LF.SOS] + hyps[b][j][::-1] + [LF.EOS]).to(hip_device).unsqueeze(-1).unsqueeze(-1)
            w1 = max(w1, float(np.abs(np.asarray(d.las_rescore(x, tgt, bw=True)) - z["bw/%d/%d" % (b, j)]).max()))
    print("one hypothesis at a time (las_rescore): max |log-prob - reference| %.2e" % w1)
    assert w1 < 1e-4, w1
