#!/usr/bin/env python
"""Times the REFERENCE's own Python modules (imported from /root/reference under the shims of oracle/pika_ref.py) on the
CPU cores of THIS host -- the CPU baseline SURVEY 8(d) asks for ("the reference's own Python modules ... on PyTorch-CPU
fp32").  bench.py runs it as a child process for its `cpu_baseline_reference` entries: in the build container from
/root/reference, on the GPU box from the staged copy under _ref_scratch/ (tools/stage_reference.py; oracle/pika_ref.py
resolves the root); only where neither exists does bench.py fall back to the round-4 constants (bench.py: CPU_REFERENCE).
PIKA_REF_THREADS / PIKA_REF_TRAIN_STEPS bound the sample.

    python tools/time_reference_cpu.py [train|decode|mbr ...]      # prints one JSON object; ~10 minutes for all three

  train : trainer/model/transducer.py:73-112 `Net.forward` (full width: 1024 / 9 TDNN + 3 transformer layers, conv-
          transformer prediction net, V = 5000; the (B,T,U,2H) concat and the dense log-softmax as written) + RNN-T loss
          (oracle/rnnt_loss_ref.c: the reference's loss is an absent third-party binding) + backward + inf-norm clip +
          Nesterov SGD, B = 2, T_in = 1000, U = 50: one warm-up step, then 2 timed steps.
  decode: decoder/transducer_decoder.py `decode_batch` + decoder/beam_transducer.py, beam 16 / n-best 16, on the
          full-width scenario of tests/decode_full_common.py (B = 4 utterances of ~2.5 s): RTF = wall / audio seconds.
  mbr   : the UNCHANGED trainer/train_transducer_mbr_bmuf_otfaug.py (:93-240: N-best decode, RNN-T backward, risk terms,
          trajectory joint, clip, SGD) on the reference's own modules through tests/golden/mbr_hooks.py, full-width model,
          B = 2 utterances of 1.5 s, beam 4: wall time from the start of its decode to the end of its optimizer step.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def time_train(threads):
    import numpy as np
    import torch
    from types import SimpleNamespace
    from oracle import pika_ref
    from oracle import rnnt as O
    O.build()
    torch.set_num_threads(threads)
    transducer = pika_ref.load_reference("trainer.model.transducer")
    B, T, U, V = 2, 1000, 50, 5000
    opt = SimpleNamespace(rnn_size=1024, local_rank=0, decoder_type="transformer", brnn=False, encoder_type="tdnn",
                          dropout=0.2, enc_layers=4, dec_layers=2, embd_dim=100, padding_idx=V)
    torch.manual_seed(777)
    model = transducer.Net(opt, 240, V)
    model.train()
    g = torch.Generator().manual_seed(5)
    data = torch.randn(B, T, 240, generator=g)
    labels = torch.randint(1, V, (B, U), generator=g)
    len_b = torch.full((B,), (T - 42 + 3) // 4, dtype=torch.int32)
    ali = torch.full((B,), U, dtype=torch.int32)
    optim = torch.optim.SGD(model.parameters(), 0.003, momentum=0.9, nesterov=True)

    def step():
        optim.zero_grad(set_to_none=True)
        out = model(data, labels, len_b, True)
        costs, grads = O.rnnt_loss(out.detach().numpy(), labels.int().numpy(), len_b.numpy(), ali.numpy(), dtype=np.float32)
        out.backward(torch.from_numpy(grads))
        torch.nn.utils.clip_grad_norm_(model.parameters(), 3.0, norm_type=float("inf"))
        optim.step()
        return float(costs.sum())
    step()
    t0 = time.perf_counter()
    n = int(os.environ.get("PIKA_REF_TRAIN_STEPS", "2"))
    for _ in range(n):
        loss = step()
    el = (time.perf_counter() - t0) / n
    return {"value": B / el, "unit": "utterances/s", "cores": threads, "kind": "reference",
            "sample": "reference transducer.Net fwd (as written) + oracle C RNN-T loss + bwd + clip + SGD, B=%d, T_in=%d, U=%d, "
                      "V=%d, warm, %d steps of %.1f s (loss %.1f)" % (B, T, U, V, n, el, loss)}


def time_decode(threads):
    import torch
    from types import SimpleNamespace
    from oracle import pika_ref
    import decode_full_common as F
    torch.set_num_threads(threads)
    transducer, tdec, beam_mod = pika_ref.load_reference("trainer.model.transducer", "decoder.transducer_decoder",
                                                         "decoder.beam_transducer")
    net = F.build(transducer, pika_ref.seeded_state_dict)
    x, x_len = F.inputs()
    args = SimpleNamespace(las_rescorer=None, las_rescorer_bw=None, bilas_rescorer=None, nonblk_reward=0.0)
    d = tdec.TransducerDecoder(net, batch_size=F.B, beam_size=F.BEAM, n_best=F.BEAM, blk=0,
                               global_scorer=beam_mod.GlobalScorer(), sm_scale=F.SM_SCALE, cuda=False, beam_prune=True,
                               args=args)
    t0 = time.perf_counter()
    with torch.no_grad():
        ret, _ = d.decode_batch(x, x_len, F.max_len(x_len))
    el = time.perf_counter() - t0
    audio = sum(F.LENS) / 100.0
    steps = max(len(h) for row in ret["predictions"] for h in row)
    return {"value": el / audio, "unit": "RTF", "cores": threads, "kind": "reference",
            "sample": "reference TransducerDecoder.decode_batch, B=%d, beam %d, n-best %d, %.1f s of audio, full-width model "
                      "(tests/decode_full_common.py), %.1f s of wall time, longest hypothesis %d symbols" % (
                          F.B, F.BEAM, F.BEAM, audio, el, steps)}


def time_mbr(threads):
    import subprocess
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "_mbr_child", str(threads)], capture_output=True,
                         text=True, timeout=3600)
    for line in out.stdout.splitlines():
        if line.startswith("MBR_TIMING "):
            return json.loads(line[len("MBR_TIMING "):])
    raise RuntimeError("MBR child failed:\n" + out.stderr[-3000:])


def _mbr_child(threads):
    """Inside a fresh process: tests/golden/mbr_hooks.py's reference mode with the full-width model and a timing hook in
    place of its gradient dump."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import mbr_hooks as M
    V = 5000
    M.V = V
    M.MODEL_ARGS = ["--encoder_type", "transformer", "--enc_layers", "4", "--decoder_type", "transformer", "--dec_layers", "2",
                    "--rnn_type", "LSTM", "--rnn_size", "1024", "--embd_dim", "100", "--dropout", "0.2",
                    "--padding_idx", str(V), "--output_dim", str(V)]
    B = 2

    def script_args(init_model, log, outdir):
        return ["--optim", "sgd", "--initial_lr", "0.001", "--final_lr", "0.001", "--num_batches_per_epoch", "2",
                "--num_epochs", "1", "--momentum", "0.9", "--block_momentum", "0.9", "--sync_period", "5", "--cuda",
                "--loader", "otf_utt", "--beam_size", "4", "--rnnt_scale", "0.1", "--sm_scale", "0.8", "--blk", "0",
                "--model_lctx", "21", "--model_rctx", "21", "--model_stride", "4", "--local_rank", "0",
                "--batch_size", str(B), "--fixture_batches", "2",
                "--init_model", init_model] + M.MODEL_ARGS + ["transducer", "unused.lst", log, outdir]
    M.script_args = script_args

    def save_model(path):
        import argparse
        import importlib
        import torch
        ap = argparse.ArgumentParser()
        for k in ("--encoder_type", "--decoder_type", "--rnn_type"):
            ap.add_argument(k)
        for k in ("--enc_layers", "--dec_layers", "--rnn_size", "--embd_dim", "--padding_idx", "--output_dim"):
            ap.add_argument(k, type=int)
        ap.add_argument("--dropout", type=float)
        opt = ap.parse_args(M.MODEL_ARGS)
        opt.local_rank, opt.brnn = 0, False
        torch.manual_seed(3)
        net = importlib.import_module("model.transducer").Net(opt, 240, V)
        with torch.no_grad():                   # a search that mixes blanks and labels (as bench.py calibrates its model)
            net.fc2.weight *= 8.0
            net.fc2.bias[0] += 6.0
        torch.save(net, path)
    M.save_seeded_model = save_model

    def timing_hook(out_path):
        import importlib
        import torch
        D = importlib.import_module("decoder.transducer_decoder")
        real_decode = D.TransducerDecoder.decode_batch
        state = {"n": 0}

        def decode_batch(self, *a, **k):
            state["t0"] = time.perf_counter()
            return real_decode(self, *a, **k)
        D.TransducerDecoder.decode_batch = decode_batch
        real_step = torch.optim.SGD.step

        def step(self, closure=None):
            r = real_step(self, closure)
            state["n"] += 1
            el = time.perf_counter() - state["t0"]
            if state["n"] == 2:             # the second batch: warm
                print("MBR_TIMING " + json.dumps({
                    "value": B / el, "unit": "utterances/s", "cores": threads, "kind": "reference",
                    "sample": "UNCHANGED train_transducer_mbr_bmuf_otfaug.py on the reference's own modules, full-width "
                              "model, B=%d utterances of 1.5 s, beam 4: decode -> optimizer step %.1f s (second batch)" % (B, el)}),
                    flush=True)
                os._exit(0)
            return r
        torch.optim.SGD.step = step
    M.install_dump_hook = timing_hook
    real_main = M.main

    import torch
    real_set = torch.set_num_threads
    torch.set_num_threads = lambda n: real_set(threads)      # mbr_hooks pins 4 threads for its golden run
    real_main("reference", "/dev/null")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "_mbr_child":
        _mbr_child(int(sys.argv[2]))
        sys.exit(0)
    which = sys.argv[1:] or ["train", "decode", "mbr"]
    threads = int(os.environ.get("PIKA_REF_THREADS", os.cpu_count() or 8))
    res = {"host": "%d of %d cores" % (threads, os.cpu_count() or 0)}
    for name in which:
        res[name] = {"train": time_train, "decode": time_decode, "mbr": time_mbr}[name](threads)
        print(name, json.dumps(res[name]), flush=True)
    print(json.dumps(res))
