#!/usr/bin/env python
"""bench.py -- headline benchmark of the RNN-T hot path on MI355X.

Workload `rnnt_loss_M1` (BASELINE.json metric): RNN-T loss fwd+bwd through
the drop-in `warp_rnnt.RNNTLoss.apply(...).sum().backward()` on a (B=32, T=1000, U=50, V=5000)
fp32 log-prob lattice, inputs resident in HBM, dense gradient produced.  One "step" = one batch
of B utterances per GPU.  N>1: one process per GPU (torch.distributed, RCCL), utterances are
independent so ranks shard them with no data-path collective (weak scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def make_inputs(B, T, U, V, dev, seed):
    """SURVEY 8d M1 inputs: log_softmax(randn) built utterance by utterance (no 2x temp)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lp = torch.empty((B, T, U + 1, V), dtype=torch.float32, device=dev)
    for n in range(B):
        lp[n].normal_(generator=g)
        lp[n] = torch.log_softmax(lp[n], dim=-1)
    gl = torch.Generator(device=dev)
    gl.manual_seed(seed + 1)
    labels = torch.randint(1, V, (B, U), generator=gl, device=dev, dtype=torch.int32)
    tl = torch.full((B,), T, dtype=torch.int32, device=dev)
    ul = torch.full((B,), U, dtype=torch.int32, device=dev)
    return lp, labels, tl, ul


def cpu_baseline(lp, labels, tl, ul, n_utts, min_seconds=10.0):
    """Reference-side CPU leg: the oracle's fp32 OpenMP port on a bounded sample of the SAME
    workload (first n_utts utterances), timed on this box's host cores.  Checker only."""
    from oracle import rnnt as O
    O.build()
    x = lp[:n_utts].cpu().numpy()
    y = labels[:n_utts].cpu().numpy()
    t_ = tl[:n_utts].cpu().numpy()
    u_ = ul[:n_utts].cpu().numpy()
    O.rnnt_loss(x[:1], y[:1], t_[:1], u_[:1], dtype=np.float32)  # page in
    done, t0 = 0, time.perf_counter()
    while True:
        costs, grads = O.rnnt_loss(x, y, t_, u_, dtype=np.float32)
        done += n_utts
        el = time.perf_counter() - t0
        if el >= min_seconds or done >= 400 * n_utts:   # ~10 s of wall time on the host cores
            break
    return {"value": done / el, "unit": "utterances/s", "cores": O.num_threads(), "kind": "port",
            "sample": "%d utterances of the same (T=%d,U=%d,V=%d) batch, oracle fp32 C/OpenMP port "
                      "(costs + dense grads), %.1f s" % (done, x.shape[1], x.shape[2] - 1,
                                                       x.shape[3], el)}, costs


def train_step_workload(args, dev, rank, world):
    """BASELINE.json configs[1] / SURVEY 8d M2: one full training step of the config-2 model fed by
    the HIP loader: pinned int16 audio (10.0 s / utterance, synthetic) -> speed/volume perturbation
    -> fbank (dither 1, as egs/fbank.conf) -> splice -> CMVN -> SpecAugment -> TDNN-Transformer encoder,
    conv-transformer prediction net, gated joint -> RNN-T loss -> backward -> inf-norm clip ->
    Nesterov SGD; with world > 1 a BMUF block sync (RCCL all-reduce of the flat parameter vector)
    every 5 steps, as in the recipe.  Lattice T' = 240."""
    from types import SimpleNamespace
    from model.transducer import Net  # drop-in import path of the training script
    from warp_rnnt import RNNTLoss
    from pika_amd.features import SpecAugment, cmvn_apply_
    from pika_amd.loader.frontend import FbankConfig, GpuFrontEnd
    B, T, U, V = args.batch, args.frames, args.labels, args.vocab
    opt = SimpleNamespace(rnn_size=1024, local_rank=0, decoder_type="transformer", brnn=False,
                          encoder_type="tdnn", dropout=0.2, enc_layers=4, dec_layers=2,
                          embd_dim=100, padding_idx=V)
    torch.manual_seed(777)      # identical initial replicas; BMUF broadcasts rank 0's anyway
    np.random.seed(777 + rank)
    model = Net(opt, 240, V).to(dev)
    model.train()
    cfg = FbankConfig(num_mel_bins=80, low_freq=40, high_freq=-200, dither=1.0, window_type="hamming")
    fe = GpuFrontEnd(cfg, dev, lctx=1, rctx=1, stride=1)
    n_samples = 400 + (T - 1) * 160      # T fbank frames at unchanged speed
    rng = np.random.default_rng(2000 + rank)
    pcms = [np.clip(rng.standard_normal(n_samples) * 3000, -32768, 32767).astype(np.int16) for _ in range(B)]
    g = torch.Generator(device=dev)
    g.manual_seed(2000 + rank)
    labels = torch.randint(1, V, (B, U), generator=g, device=dev)
    ali = torch.full((B,), U, dtype=torch.int32, device=dev)
    offset = torch.full((240,), -8.0, device=dev)
    scale = torch.full((240,), 0.25, device=dev)
    loss_fn = RNNTLoss(blank=0, reduction="sum").apply
    aug = SpecAugment(15, 35)
    state = {"optim": torch.optim.SGD(model.parameters(), 0.003, momentum=0.9, nesterov=True), "n": 0}
    bmuf = None
    if world > 1:
        from trainer.bmuf import BmufTrainer
        bmuf = BmufTrainer(0, rank, world, model, 0.9, 1.0)

    def step():
        state["optim"].zero_grad(set_to_none=True)
        # speed 1.0 keeps the benchmark shape fixed (T frames); the level perturbation is drawn
        dbs = np.random.uniform(-50.0, -10.0, B)
        data, flens = fe(pcms, [1.0] * B, list(dbs))
        lens = torch.tensor(flens, dtype=torch.int32, device=dev)
        len_b = lens - 42                       # train_transducer_bmuf_otfaug.py:80-82
        len_b = len_b // 4 + (len_b % 4 != 0).int()
        cmvn_apply_(data, offset, scale, cmn=True)
        aug.apply(data)
        out = model(data, labels, len_b, True)
        loss = loss_fn(out, labels.int(), len_b, ali).sum()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 3.0, norm_type=float("inf"))
        state["optim"].step()
        state["n"] += 1
        if bmuf is not None and state["n"] % 5 == 0:      # sync_period 5 (:112-123)
            assert bmuf.update_and_sync()
            state["optim"] = torch.optim.SGD(model.parameters(), 0.003, momentum=0.9, nesterov=True)
        return loss

    flops_per_utt = 730e9  # ~243 GF fwd, x3 fwd+bwd (split fc1/fc_gate)
    return step, flops_per_utt


def run_train_step(args, dev, rank, world, steps, warmup):
    from pika_amd import gemm as G
    B, T, U, V = args.batch, args.frames, args.labels, args.vocab
    step, flops_per_utt = train_step_workload(args, dev, rank, world)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    tf = flops_per_utt * B / (el / steps) / 1e12
    return {"metric": "utterances/sec RNNT train step (T_in=%d,U=%d,V=%d)" % (T, U, V),
            "value": B * world / (el / steps), "unit": "utterances/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if G.PRECISION == "bf16" else "f32-split", "data": "synthetic",
            "config": {"workload": "train_step (BASELINE configs[1]): full PIKA TDNN-Transformer RNN-T, HIP "
                                   "loader from pinned int16 audio (fbank+splice), CMVN, SpecAugment, fwd, "
                                   "RNN-T loss, bwd, clip, SGD%s" % (", BMUF all-reduce every 5 steps" if world > 1 else ""),
                       "batch_per_gpu": B, "T_in": T, "T_enc": 240, "U": U, "V": V,
                       "global_batch": B * world, "parallelism": "bmuf-dp%d" % world,
                       "loss": float(loss.item())},
            "roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": tf / 2500.0, "traffic": None}}


def decode_workload(args, dev, rank):
    """batch beam decode, B utterances x beam 16, 10 s of synthetic fbank each,
    full-size model with random weights.  Random weights never emit blank, so fc2 is sharpened
    and the blank bias calibrated (greedy, 8 utterances) until an utterance emits ~U=50 labels
    over its T'=240 frames -- the step count (T'+U) of a trained model."""
    from types import SimpleNamespace
    from model.transducer import Net
    from decoder.transducer_decoder import TransducerDecoder
    from decoder.beam_transducer import GlobalScorer
    B, T, V = args.batch, args.frames, args.vocab
    opt = SimpleNamespace(rnn_size=1024, local_rank=0, decoder_type=args.pred_net, brnn=False,
                          encoder_type="tdnn", dropout=0.2, enc_layers=4, dec_layers=2,
                          embd_dim=100, padding_idx=V)
    torch.manual_seed(777 + rank)
    model = Net(opt, 240, V).to(dev).eval()
    g = torch.Generator(device=dev)
    g.manual_seed(3000 + rank)
    feats = (torch.randn(B, T, 240, generator=g, device=dev)).contiguous()
    x_len = torch.full((B,), (T - 42 + 3) // 4, dtype=torch.long, device=dev)
    dargs = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=0.0)
    lm_scorer = synthetic_bigram_matcher(V) if args.fst else None
    las_fw = las_bw = None
    SOS, EOS, PAD = V, V + 1, V + 2
    if args.las:   # forward + backward LAS rescorers, 2-layer BLSTM 1024, mlp attention, random weights
        from trainer.model import las
        lopt = SimpleNamespace(rnn_size=1024, encoder_type="rnn", rnn_type="LSTM", brnn=True, enc_layers=2,
                               dropout=0.0, use_downsampler=False, embd_dim=100, num_heads=1, sampling_decoder=False,
                               input_feed=1, dec_layers=2, global_attention="mlp", coverage_attn=False,
                               context_gate=None, copy_attn=False)
        torch.manual_seed(999)
        las_fw = las.Net(lopt, 1024, V + 2, PAD).to(dev).eval()
        las_bw = las.Net(lopt, 1024, V + 2, PAD).to(dev).eval()

    def decoder(beam, nbest, lm=None):
        return TransducerDecoder(model, batch_size=B, beam_size=beam, n_best=nbest, blk=0,
                                 global_scorer=GlobalScorer(), sm_scale=0.8, cuda=True,
                                 lm_scorer=lm, lm_scorer_scale=args.fst_scale,
                                 beam_prune=True, args=dargs)

    with torch.no_grad():
        model.fc2.weight *= 8.0
        lo, hi = 0.0, 40.0
        labels = float("nan")
        if args.blank_bias is not None:      # profiling runs: skip the calibration decodes
            lo = hi = args.blank_bias
        for _ in range(8 if args.blank_bias is None else 0):  # bisection on the blank bias
            mid = 0.5 * (lo + hi)
            model.fc2.bias[0] = mid
            ret, _ = decoder(1, 1).decode_batch(feats[:8], x_len[:8], [int(v) + 100 for v in x_len[:8]])
            labels = np.mean([sum(1 for e in h[0] if int(e) != 0) for h in ret["predictions"]])
            if labels > args.labels:
                lo = mid
            else:
                hi = mid
        model.fc2.bias[0] = 0.5 * (lo + hi)
        decode_workload.blank_bias = 0.5 * (lo + hi)
    dec = decoder(args.beam, args.beam, lm_scorer)

    def step():
        ret, enc_out = dec.decode_batch(feats, x_len, [int(v) + 100 for v in x_len])
        t0 = time.perf_counter()
        if las_fw is not None:      # decode_transducer.py:136-156: every n-best entry, forward and reversed
            hyps = [[[int(e) for e in h if int(e) != 0] for h in ret["predictions"][i]] for i in range(B)]
            src = enc_out.transpose(0, 1)                                   # (T', B, H)
            fw = las_fw.score_nbest_batch(src, x_len, hyps, SOS, EOS)
            bw = las_bw.score_nbest_batch(src, x_len, [[h[::-1] for h in row] for row in hyps], SOS, EOS)
            ret["las"] = (fw, bw)
            torch.cuda.synchronize()
        dec.timing["las_s"] = time.perf_counter() - t0
        return ret, enc_out
    step.decoder = dec
    return step, float(labels)


def synthetic_bigram_matcher(V, seed=123, n_succ=6):
    """seeded back-off bigram LM over the V-1 labels as an FST (word w has ilabel w+1, back-off
    label V+1, one disambiguation label V+2), wrapped in the matcher the fused search queries."""
    from pika_amd.decoder.ngram_fst import NgramFst, SortedMatcher
    rng = np.random.default_rng(seed)
    backoff_id, disambig = V + 1, V + 2
    arcs, finals = [], {0: float(np.float32(rng.uniform(1, 3)))}
    for w in range(1, V):
        arcs.append((0, w + 1, float(np.float32(rng.uniform(2, 7))), 1 + w))
    for w in range(1, V):
        h = 1 + w
        for v in rng.choice(np.arange(1, V), size=n_succ, replace=False):
            arcs.append((h, int(v) + 1, float(np.float32(rng.uniform(0.5, 4))), 1 + int(v)))
        arcs.append((h, backoff_id, float(np.float32(rng.uniform(0.2, 2))), 0))
        if w % 3 == 0:
            finals[h] = float(np.float32(rng.uniform(0.5, 2)))
    return SortedMatcher(NgramFst.from_arcs(1 + V, arcs, finals), max_num_arcs=V + 4, max_id=V + 3,
                         backoff_id=backoff_id, disambig_ids=[disambig])


def mbr_workload(args, dev, rank):
    """SURVEY 8d M4 (BASELINE configs[3], one GPU): one minimum-Bayes-risk training step as
    train_transducer_mbr_bmuf_otfaug.py:112-240 runs it -- N-best beam decode (eval mode, beam 4) -> encoder
    forward -> RNN-T loss on the reference labels, backward (retain graph) -> risk terms (softmax of the N-best
    scores, edit distances) -> prediction net on the N-best label sequences -> joint along every hypothesis'
    (t,u) trajectory with the HIP risk-gradient kernel -> inf-norm clip -> Nesterov SGD.  Random weights never
    emit blank, so fc2 is sharpened and the blank bias calibrated as in the decode workload."""
    from types import SimpleNamespace
    from model.transducer import Net
    from warp_rnnt import RNNTLoss
    from decoder.transducer_decoder import TransducerDecoder
    from decoder.beam_transducer import GlobalScorer
    from pika_amd import mbr
    from pika_amd.model import ops
    B, T, U, V, beam = args.batch, args.frames, args.labels, args.vocab, args.beam
    opt = SimpleNamespace(rnn_size=1024, local_rank=0, decoder_type="transformer", brnn=False,
                          encoder_type="tdnn", dropout=0.2, enc_layers=4, dec_layers=2,
                          embd_dim=100, padding_idx=V)
    torch.manual_seed(777 + rank)
    model = Net(opt, 240, V).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(4000 + rank)
    feats = torch.randn(B, T, 240, generator=g, device=dev).contiguous()
    x_len = torch.full((B,), (T - 42 + 3) // 4, dtype=torch.long, device=dev)
    labels = torch.randint(1, V, (B, U), generator=g, device=dev)
    ali = torch.full((B,), U, dtype=torch.int32, device=dev)
    dargs = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=0.0)

    def decoder(k):
        return TransducerDecoder(model, batch_size=B, beam_size=k, n_best=k, blk=0, global_scorer=GlobalScorer(),
                                 sm_scale=0.8, cuda=True, beam_prune=True, args=dargs)
    model.eval()
    with torch.no_grad():
        model.fc2.weight *= 8.0
        lo, hi = 0.0, 40.0
        for _ in range(8):
            mid = 0.5 * (lo + hi)
            model.fc2.bias[0] = mid
            ret, _ = decoder(1).decode_batch(feats, x_len, [int(v) + 100 for v in x_len])
            nlab = np.mean([sum(1 for e in h[0] if int(e) != 0) for h in ret["predictions"]])
            lo, hi = (mid, hi) if nlab > U else (lo, mid)
        model.fc2.bias[0] = 0.5 * (lo + hi)
    for m in model.modules():   # keep the eval-mode model (running statistics) at its calibration point
        if isinstance(m, torch.nn.BatchNorm1d):
            m.momentum = 0.0
    dec = decoder(beam)
    # the step size is negligible on purpose: the calibrated random model must keep emitting ~U labels per
    # utterance in every timed step (the arithmetic of the update is the same)
    optim = torch.optim.SGD(model.parameters(), 1e-9, momentum=0.9, nesterov=True)
    loss_fn = RNNTLoss(blank=0, reduction="sum").apply
    max_len = [int(v) + U + 3 for v in x_len]
    info = {}

    def step():
        model.eval()
        with torch.no_grad():
            ret, _ = dec.decode_batch(feats, x_len, max_len)                  # :112-117
        hyps, scores = ret["predictions"], ret["scores"]
        model.train()
        optim.zero_grad(set_to_none=True)
        enc = model.encode(feats, None)                                       # :124-138
        sos = torch.zeros(B, 1, dtype=torch.long, device=dev)
        pred = model.predict(torch.cat((sos, labels), dim=1))
        lp = ops.joint(enc, pred, model.fc1, model.fc_gate, model.fc2, log_softmax=True)
        rnnt = 0.1 * loss_fn(lp, labels.int(), x_len.int(), ali).sum()       # rnnt_scale :152-158
        rnnt.backward(retain_graph=True)
        prob, dist, seq_grad, nonblk = mbr.risk_terms(hyps, scores, labels, ali, 0, dev)   # :163-195
        mbr.mbr_backward(model, enc, hyps, seq_grad, nonblk, 0, 0.8)          # :197-235
        torch.nn.utils.clip_grad_norm_(model.parameters(), 3.0, norm_type=float("inf"))
        optim.step()
        info["risk"] = float((prob * dist).sum())
        info["hyp_labels"] = float(np.mean([len(h) for row in nonblk for h in row]))
        return rnnt
    return step, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="rnnt_loss_M1", choices=["rnnt_loss_M1", "rnnt_loss_M1p", "train_step", "decode", "mbr_step"])
    ap.add_argument("--beam", type=int, default=16)
    ap.add_argument("--pred-net", default="transformer", choices=["transformer", "rnn"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--labels", type=int, default=50)
    ap.add_argument("--vocab", type=int, default=5000)
    ap.add_argument("--cpu-utts", type=int, default=4)
    ap.add_argument("--blank-bias", type=float, default=None,
                    help="decode: use this fc2 blank bias instead of calibrating it (profiling runs)")
    ap.add_argument("--fst", action="store_true", help="decode: n-gram FST shallow fusion (synthetic bigram)")
    ap.add_argument("--fst-scale", type=float, default=0.02,
                    help="decode --fst: LM weight (small: the synthetic LM is random, the search should keep emitting)")
    ap.add_argument("--las", action="store_true", help="decode: forward + backward LAS rescoring of the n-best")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true",
                    help="skip the secondary full-train-step measurement of the default run")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    if "PIKA_BENCH_DEVICE" in os.environ:      # test hook: several ranks on one GPU (gloo backend)
        local_rank = int(os.environ["PIKA_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=os.environ.get("PIKA_BENCH_BACKEND", "nccl"), init_method="env://")

    from warp_rnnt import RNNTLoss  # the drop-in import the reference scripts use
    from pika_amd import rnnt as R

    B, T, U, V = args.batch, args.frames, args.labels, args.vocab
    if args.workload == "decode":
        step, cal_labels = decode_workload(args, dev, rank)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ret, _ = step()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.steps
        audio_s = B * T / 100.0
        hyps = ret["predictions"]
        nlab = float(np.mean([sum(1 for e in h[0] if int(e) != 0) for h in hyps]))
        nsteps = float(np.mean([len(h[0]) + 1 for h in hyps]))
        if rank == 0:
            print(json.dumps({
                "metric": "decode RTF (wall / audio seconds), batch beam search", "value": el / audio_s,
                "unit": "RTF", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": el * 1e3, "higher_is_better": False, "scaling": "weak",
                "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": "decode: B=%d beam=%d n_best=%d, %d-frame utterances, full model "
                                       "(%s prediction net), sm_scale 0.8%s%s" % (
                                           B, args.beam, args.beam, T, args.pred_net,
                                           ", bigram FST shallow fusion (device-resident FST, scale %g)" % args.fst_scale if args.fst else "",
                                           ", fw+bw LAS rescoring of the n-best" if args.las else ""),
                           "audio_seconds": audio_s, "utterances_per_s": B / el,
                           "labels_per_utt_top1": nlab, "search_steps_top1": nsteps,
                           "calibration_labels": cal_labels, "blank_bias": decode_workload.blank_bias,
                           "timing": step.decoder.timing}}), flush=True)
        return
    if args.workload == "rnnt_loss_M1p":
        # SURVEY 8d M1': fused boundary logits -> (costs, d/dlogits); no log-prob tensor, no dense lp gradient
        from pika_amd.rnnt import rnnt_loss_from_logits
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + 100 * rank)
        logits = torch.randn(B, T, U + 1, V, generator=g, device=dev).requires_grad_(True)
        g.manual_seed(1235 + 100 * rank)
        labels = torch.randint(1, V, (B, U), generator=g, device=dev, dtype=torch.int32)
        tl = torch.full((B,), T, dtype=torch.int32, device=dev)
        ul = torch.full((B,), U, dtype=torch.int32, device=dev)

        def step():
            logits.grad = None
            c = rnnt_loss_from_logits(logits, labels, tl, ul)
            c.sum().backward()
            return c
        for _ in range(args.warmup):
            step()
        R.KERNEL_EVENTS = {"fwd": [], "bwd": []}
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            costs = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        ev, R.KERNEL_EVENTS = R.KERNEL_EVENTS, None
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item()) / args.steps
        if rank == 0:
            fwd_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["fwd"]]))
            bwd_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["bwd"]]))
            bytes_per_launch = 3.0 * B * T * (U + 1) * V * 4          # SURVEY 8d M1': 3X
            achieved = bytes_per_launch / ((fwd_ms + bwd_ms) * 1e-3) / 1e9
            # the composition it replaces, on the same logits
            lp = torch.log_softmax(logits.detach(), dim=-1)
            c_ref = RNNTLoss(blank=0).apply(lp, labels, tl, ul)
            print(json.dumps({
                "metric": "utterances/sec fused log-softmax + RNNT loss fwd+bwd (T=%d,U=%d,V=%d)" % (T, U, V),
                "value": B * world / el, "unit": "utterances/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": el * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "rnnt_loss_M1p (SURVEY 8d M1'): rnnt_loss_from_logits(randn logits (B,T,U+1,V))"
                                       ".sum().backward(), fp32 d/dlogits out", "batch_per_gpu": B, "T": T, "U": U, "V": V,
                           "max_rel_cost_diff_vs_log_softmax_plus_loss": float(((costs - c_ref).abs() / c_ref.abs()).max())},
                "roofline": {"bound": "hbm", "kernel": "rnnt_lse_gather_kernel + rnnt_dlogits_fused_kernel",
                             "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "bytes_per_launch": bytes_per_launch,
                             "forward_ms": fwd_ms, "backward_ms": bwd_ms}}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.workload == "mbr_step":
        step, info = mbr_workload(args, dev, rank)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item()) / args.steps
        if rank == 0:
            print(json.dumps({
                "metric": "utterances/sec MBR train step (T_in=%d,U=%d,V=%d)" % (T, U, V), "value": B * world / el,
                "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": el * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16", "data": "synthetic",
                "config": {"workload": "mbr_step (BASELINE configs[3] / SURVEY 8d M4): N-best decode (beam %d) + encoder "
                                       "fwd + RNN-T loss bwd + risk terms + trajectory joint with the HIP risk-gradient "
                                       "kernel + clip + SGD, full config-2 model" % args.beam,
                           "batch_per_gpu": B, "beam": args.beam, "expected_risk": info.get("risk"),
                           "hyp_labels": info.get("hyp_labels")}}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.workload == "train_step":
        out = run_train_step(args, dev, rank, world, args.steps, args.warmup)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    lp, labels, tl, ul = make_inputs(B, T, U, V, dev, 1234 + 100 * rank)
    lp.requires_grad_(True)
    loss_fn = RNNTLoss(blank=0, reduction="sum").apply

    def step():
        lp.grad = None
        costs = loss_fn(lp, labels, tl, ul)
        costs.sum().backward()
        return costs

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    R.KERNEL_EVENTS = {"fwd": [], "bwd": []}  # HIP events on the launch stream, per C-ABI call
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        costs = step()
    barrier()
    el = time.perf_counter() - t0
    ev = R.KERNEL_EVENTS
    R.KERNEL_EVENTS = None
    t = torch.tensor([el], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())

    if rank == 0:
        ms_step = el / args.steps * 1e3
        value = B * world / (el / args.steps)
        bwd_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["bwd"]]))
        fwd_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["fwd"]]))
        cells = T * (U + 1)
        bytes_per_utt = cells * V * 4 + 2 * cells * 4 + 4 * cells * 4
        achieved = bytes_per_utt * B / (bwd_ms * 1e-3) / 1e9
        # measured ceiling for a pure write stream of the same size (hipMemsetAsync)
        gbuf = lp.grad
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            gbuf.zero_()
        e1.record()
        torch.cuda.synchronize()
        fill_gbps = gbuf.numel() * 4 * 3 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "rnnt_grad_pmc.json")
        if os.path.exists(pmc):
            # measured at (B=32,T=1000,U=50,V=5000); bytes scale with the batch, valid for that lattice only
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch") * B / 32.0 \
                if (T, U, V) == (1000, 50, 5000) else None
        out = {
            "metric": "utterances/sec RNNT fwd+bwd (T=%d,U=%d,V=%d)" % (T, U, V),
            "value": value, "unit": "utterances/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "rnnt_loss_M1: warp_rnnt.RNNTLoss.apply(...).sum().backward() on "
                                   "log_softmax(randn) (B,T,U+1,V) fp32, dense grad out",
                       "batch_per_gpu": B, "T": T, "U": U, "V": V, "global_batch": B * world,
                       "parallelism": "utterance-sharded x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "kernel": "rnnt_grad_kernel", "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "bytes_per_launch": bytes_per_utt * B,
                         "kernel_ms": bwd_ms, "forward_ms": fwd_ms,
                         "measured_fill_GBps": fill_gbps},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, c_cpu = cpu_baseline(lp.detach(), labels, tl, ul, min(args.cpu_utts, B))
            out["cpu_baseline"] = cb
            c_gpu = costs.detach()[:len(c_cpu)].cpu().numpy()
            out["cpu_baseline"]["max_rel_cost_diff_vs_gpu"] = float(
                np.max(np.abs(c_gpu - c_cpu) / np.abs(c_cpu)))
    if not args.no_train_step:
        # secondary measurement in the same run: the full configs[1] training step
        lp.grad = None
        gbuf = costs = None  # noqa: F841  (drop the 2 x 32 GB of the loss workload)
        del lp, labels, tl, ul
        torch.cuda.empty_cache()
        try:
            ts = run_train_step(args, dev, rank, world, max(5, min(args.steps, 10)), 2)
            ts = {k: ts[k] for k in ("value", "unit", "ms_per_step", "dtype", "config", "roofline")}
        except Exception as e:  # the headline line must survive a failure of the secondary leg
            ts = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0:
            out["train_step"] = ts
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
