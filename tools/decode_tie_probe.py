#!/usr/bin/env python
"""Which product moves the beam-boundary ties?  (VERDICT r5 next #4.)  The full-width n-best search of
tests/test_decode_full.py on the GPU with each product family switched to exact fp32 products in turn -- the encoder pass +
joint halves (`encoder_precision`: "fp16x2" K-concatenated two-fp16-term products vs "fp32" six-segment exact), the step
products (`decode_precision`: "fp32" two fp16 terms vs "fp32-exact" three bf16 terms) -- against the float64 run of the
reference decoder (tests/golden/decode_full_f64.npz): entries the float64 search finished too, entries in float64 order,
and WHICH entries are missing per utterance.       python tools/decode_tie_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import test_decode_full as T
    from types import SimpleNamespace
    z = np.load(T.GOLD)
    f = np.load(T.GOLD_F64)
    dev = torch.device("cuda:0")
    B, nb = z["lens"].shape
    print("fp32 REFERENCE list: ", end="")
    for b in range(B):
        r, e = T.against_f64(z, f, b)
        print("utt %d missing at ranks %s (err %.1e); " % (b, [j for j, k in enumerate(r) if k < 0], e), end="")
    print()
    combos = [("fp16x2", "fp32"), ("fp32", "fp32"), ("fp16x2", "fp32-exact"), ("fp32", "fp32-exact")]
    extra = os.environ.get("TIE_PROBE_EXTRA", "")
    for enc_p, step_p in combos:
        sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
        from decoder.transducer_decoder import TransducerDecoder
        from decoder.beam_transducer import GlobalScorer
        from pika_amd.model import transducer
        import decode_common as D
        import decode_full_common as F
        from oracle.pika_ref import seeded_state_dict
        net = F.build(transducer, seeded_state_dict).to(dev)
        x, x_len = F.inputs()
        args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=0.0)
        d = TransducerDecoder(net, batch_size=F.B, beam_size=F.BEAM, n_best=F.BEAM, blk=0, global_scorer=GlobalScorer(),
                              sm_scale=F.SM_SCALE, cuda=True, beam_prune=True, args=args)
        d.decode_precision, d.encoder_precision = step_p, enc_p
        ret, enc = d.decode_batch(x.to(dev), x_len.to(dev), F.max_len(x_len))
        got = D.pack(ret["predictions"], ret["scores"])
        found = inorder = 0
        worst = 0.0
        detail = []
        for b in range(B):
            r, e = T.against_f64(got, f, b)
            ks = [k for k in r if k >= 0]
            found += len(ks)
            inorder += sum(int(k == o) for k, o in zip(ks, sorted(ks)))
            worst = max(worst, e)
            miss = [j for j, k in enumerate(r) if k < 0]
            if miss:
                # the float64 entries the GPU list lacks, with the float64 score gap to the float64 list's neighbours
                have = set(ks)
                lack = [k for k in range(min(nb, int(f["count"][b]))) if k not in have]
                detail.append("utt %d: GPU ranks %s not in the f64 list; f64 ranks %s not in the GPU list (f64 scores %s)" % (
                    b, miss, lack, ["%.4f" % f["scores"][b, k] for k in lack]))
        es = enc[:, ::7, ::37].float().cpu().numpy()
        rel = np.abs(es - z["enc_sample"]).max() / np.abs(z["enc_sample"]).max()
        print("encoder %-7s step %-10s: %d of %d entries in the float64 list, %d in float64 order, max |score - f64| %.2e, "
              "encoder output %.1e off the fp32 reference's" % (enc_p, step_p, found, B * nb, inorder, worst, rel))
        for s in detail:
            print("    " + s)
        del d, net
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
