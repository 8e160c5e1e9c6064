#!/usr/bin/env python
"""Runs the UNCHANGED reference scripts on the MI355X through `python -m pika_amd.launch` with NO preload:
trainer/train_transducer_bmuf_otfaug.py (1 epoch x 20 batches of a synthetic corpus, full-width model, HIP loader,
HIP RNN-T loss, BMUF with the RCCL backend) and then decoder/decode_transducer.py on the checkpoint it wrote
(command line of egs/eval_transducer.sh:74-101 with forward + backward LAS rescorers -- randomly initialised 2-layer BLSTM
1024 / mlp-attention models pickled like trained ones -- and without the optional FST LM).

The scripts are NOT part of this repository: `stage` copies the two files from /root/reference into the git-ignored
scratch directory _ref_scratch/ (so that they travel with the gpurun snapshot), `clean` removes it again.

    python tools/ref_scripts_gpu.py stage                      # in the build container
    gpurun -- 'python tools/ref_scripts_gpu.py run gpurun_out/r2_scripts'
    python tools/ref_scripts_gpu.py clean
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRATCH = os.path.join(ROOT, "_ref_scratch")
FILES = ["trainer/train_transducer_bmuf_otfaug.py", "decoder/decode_transducer.py"]


def stage():
    for f in FILES:
        dst = os.path.join(SCRATCH, f)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join("/root/reference", f), dst)
    print("staged", FILES, "->", SCRATCH)


def clean():
    shutil.rmtree(SCRATCH, ignore_errors=True)


def run(outdir):
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from pathlib import Path
    from test_loader import make_corpus
    from pika_amd.loader import kaldi_io
    out = Path(outdir).resolve()
    out.mkdir(parents=True, exist_ok=True)
    work = out / "work"
    work.mkdir(exist_ok=True)
    V = 5000
    # 20 batches x 8 utterances of 2.5-4 s; labels are drawn below 50 by make_corpus (a sub-range of the vocabulary)
    lst, conf, _, _ = make_corpus(work, n_utts=160, seed=41, lo=40000, hi=64000)
    (work / "fbank.conf").write_text(Path(conf).read_text().replace("--dither=0", "--dither=1"))   # egs/fbank.conf
    rng = np.random.default_rng(0)
    n, mean = 1000.0, rng.normal(8, 1, 80)
    (work / "cmvn.stats").write_text(" [\n  " + " ".join("%.10g" % v for v in np.concatenate((mean * n, [n]))) + "\n  " +
                                     " ".join("%.10g" % v for v in np.concatenate(((mean ** 2 + 4.0) * n, [0.0]))) + " ]\n")
    (work / "ckpt").mkdir(exist_ok=True)
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517",
               PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    train = [sys.executable, "-m", "pika_amd.launch", os.path.join(SCRATCH, FILES[0]),
             "--verbose", "--optim", "sgd", "--initial_lr", "0.003", "--final_lr", "0.0001", "--grad_clip", "3.0",
             "--num_batches_per_epoch", "20", "--num_epochs", "1", "--momentum", "0.9", "--block_momentum", "0.9",
             "--sync_period", "5", "--feats_dim", "80", "--cuda", "--batch_size", "8", "--encoder_type", "transformer",
             "--enc_layers", "4", "--decoder_type", "transformer", "--dec_layers", "2", "--rnn_type", "LSTM",
             "--rnn_size", "1024", "--embd_dim", "100", "--dropout", "0.2", "--padding_idx", str(V), "--padding_tgt", str(V),
             "--stride", "1", "--queue_size", "8", "--loader", "otf_utt", "--batch_first", "--cmn",
             "--cmvn_stats", str(work / "cmvn.stats"), "--output_dim", str(V), "--num_workers", "1",
             "--sample_rate", "16000", "--feat_config", str(work / "fbank.conf"), "--TU_limit", "15000",
             "--gain_range", "50,10", "--speed_rate", "0.9,1.0,1.1", "--spec_augment", "--log_per_n_frames", "200",
             "--max_len", "1600", "--lctx", "1", "--rctx", "1", "--model_lctx", "21", "--model_rctx", "21",
             "--model_stride", "4", "--local-rank=0", "transducer", lst, str(out / "train.WORKER-ID.log"),
             str(work / "ckpt")]
    prof = ["rocprofv3", "--kernel-trace", "--stats", "-d", str(out / "prof_train"), "-o", "train", "--"]
    r = subprocess.run(prof + train, env=env, cwd=str(work), capture_output=True, text=True)
    (out / "train.stdout").write_text(r.stdout[-4000:])
    (out / "train.stderr").write_text(r.stderr[-6000:])
    print("train rc", r.returncode)
    model = work / "ckpt" / "model.epoch.0.0"
    assert r.returncode == 0 and model.exists(), r.stderr[-2000:]
    # decode: Kaldi archives of 80-d features + labels (utt loader), the checkpoint the training script wrote
    utts = [("utt%d" % i, rng.normal(0, 1, (int(k), 80)).astype(np.float32)) for i, k in enumerate(rng.integers(250, 400, 16))]
    kaldi_io.write_matrix_ark(str(work / "feats.ark"), utts)
    kaldi_io.write_int_vectors(str(work / "labels.ark"), [(k, np.array([1, 2, 3])) for k, _ in utts], binary=False)
    (work / "sym.map").write_text("".join("s%d %d\n" % (i, i) for i in range(V + 1)))
    # forward / backward LAS rescorers as whole-module pickles (what train_las_bmuf_otfaug.py writes)
    import torch
    from types import SimpleNamespace
    sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
    from model import las
    lopt = SimpleNamespace(rnn_size=1024, encoder_type="rnn", rnn_type="LSTM", brnn=True, enc_layers=2, dropout=0.0,
                           use_downsampler=False, embd_dim=100, num_heads=1, sampling_decoder=False, input_feed=1,
                           dec_layers=2, global_attention="mlp", coverage_attn=False, context_gate=None, copy_attn=False)
    for seed, name in ((3, "las.mdl"), (4, "las_bw.mdl")):
        torch.manual_seed(seed)
        torch.save(las.Net(lopt, 1024, V + 1, V + 1), str(work / name))
    decode = [sys.executable, "-m", "pika_amd.launch", os.path.join(SCRATCH, FILES[1]),
              "--verbose", "--cuda", "--min_len", "50", "--blk", "0", "--batch_first", "--beam_size", "8", "--output_scores",
              "--sm_scale", "0.8", "--batch_size", "8", "--n_best", "4", "--SOS", "0", "--EOS", str(V), "--padding_idx", str(V),
              "--loader", "utt", "--lctx", "1", "--rctx", "1", "--feats_dim", "80", "--model_lctx", "21", "--model_rctx", "21",
              "--model_stride", "4", "--las_rescorer_model", str(work / "las.mdl"), "--las_rescorer_bw_model", str(work / "las_bw.mdl"),
              "--symbols_map", str(work / "sym.map"), str(model),
              "ark:" + str(work / "feats.ark"), "ark:" + str(work / "labels.ark"), str(out / "hyp.txt")]
    prof = ["rocprofv3", "--kernel-trace", "--stats", "-d", str(out / "prof_decode"), "-o", "decode", "--"]
    r = subprocess.run(prof + decode, env=env, cwd=str(work), capture_output=True, text=True)
    (out / "decode.stderr").write_text(r.stderr[-6000:])
    # like the reference, the script ends on the `None` its loader yields last (decode_transducer.py:108)
    print("decode rc", r.returncode, "n-best lines", len((out / "hyp.txt").read_text().splitlines()))
    for name in ("train", "decode"):
        dbs = list((out / ("prof_" + name)).rglob("*_results.db"))
        if dbs:
            txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_stats.py"), str(dbs[0]), "--top", "120"],
                                 capture_output=True, text=True).stdout
            (out / ("%s_kernel_stats.csv" % name)).write_text(txt)
            for d in dbs:
                d.unlink()          # the databases are large; the summary is what is kept
    shutil.rmtree(work, ignore_errors=True)


def train_cost(outdir):
    """Marginal cost of a training step of the UNCHANGED script: the same command on a corpus of 40 and of 200 batches,
    process CPU time (user + sys, all threads: training loop + loader) and wall time of the child; (run2 - run1) / 160 steps is
    what a step costs once the graphs exist.  The corpus is length-bucketed the way the recipes' split_by_length.py makes
    a rank's batches (utterances of 3.25-3.6 s, speed perturbation 0.9 / 1.0 / 1.1: batch frame counts within a few tens of
    frames of each other), which the graphs' padded time / label axes absorb; PIKA_TRAIN_GRAPH=0 gives the eager launch sequence on the same corpus."""
    import json
    import resource
    import time
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from pathlib import Path
    from test_loader import make_corpus
    out = Path(outdir).resolve()
    out.mkdir(parents=True, exist_ok=True)
    work = out / "work"
    work.mkdir(exist_ok=True)
    V = 5000
    # (an epoch of the script is one pass over its list, whatever --num_batches_per_epoch says: the run length is the corpus)
    corpora = {}
    for nb in [int(v) for v in os.environ.get("COST_BATCHES", "40,200").split(",")]:
        d = work / ("corpus%d" % nb)
        d.mkdir(exist_ok=True)
        # utterances of 3.25-3.6 s (x 0.9 / 1.0 / 1.1 speed perturbation): the longest utterance of a batch of 8 -- the
        # batch's frame count -- varies over ~20 frames, which the padded time axis of the graphs absorbs
        corpora[nb] = make_corpus(d, n_utts=8 * nb, seed=41, lo=52000, hi=57600)[:2]
    conf = corpora[min(corpora)][1]
    (work / "fbank.conf").write_text(Path(conf).read_text().replace("--dither=0", "--dither=1"))
    rng = np.random.default_rng(0)
    n, mean = 1000.0, rng.normal(8, 1, 80)
    (work / "cmvn.stats").write_text(" [\n  " + " ".join("%.10g" % v for v in np.concatenate((mean * n, [n]))) + "\n  " +
                                     " ".join("%.10g" % v for v in np.concatenate(((mean ** 2 + 4.0) * n, [0.0]))) + " ]\n")
    res = {}
    modes = (("graphs", {"PIKA_TRAIN_GRAPH_DEBUG": "1"}), ("eager", {"PIKA_TRAIN_GRAPH": "0"}))
    if os.environ.get("COST_VARIANTS"):     # debugging: name=ENV=VALUE,... ; e.g. "a=PIKA_TRAIN_GRAPH_U_BUCKET=1;b=COST_DROPOUT=0"
        modes = []
        for item in os.environ["COST_VARIANTS"].split(";"):
            name, _, kv = item.partition("=")
            e = {"PIKA_TRAIN_GRAPH_DEBUG": "1"}
            for pair in filter(None, kv.split(",")):
                k, _, v = pair.partition("=")
                e[k] = v
            modes.append((name, e))
    if not os.environ.get("COST_VARIANTS"):
        modes = (("coldstart", {"PIKA_TRAIN_GRAPH": "0"}),) + tuple(modes)      # the first process pages the image in
    for mode, genv in modes:
        rows = []
        for nb in sorted(corpora):
            lst = corpora[nb][0]
            ck = work / ("ckpt_%s_%d" % (mode, nb))
            ck.mkdir(exist_ok=True)
            env = dict(os.environ, WORLD_SIZE="1", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517",
                       PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", **genv)
            train = [sys.executable, "-m", "pika_amd.launch", os.path.join(SCRATCH, FILES[0]),
                     "--verbose", "--optim", "sgd", "--initial_lr", "0.003", "--final_lr", "0.0001", "--grad_clip", "3.0",
                     "--num_batches_per_epoch", str(nb), "--num_epochs", "1", "--momentum", "0.9", "--block_momentum", "0.9",
                     "--sync_period", "5", "--feats_dim", "80", "--cuda", "--batch_size", "8", "--encoder_type", "transformer",
                     "--enc_layers", "4", "--decoder_type", "transformer", "--dec_layers", "2", "--rnn_type", "LSTM",
                     "--rnn_size", "1024", "--embd_dim", "100", "--dropout", genv.get("COST_DROPOUT", "0.2"), "--padding_idx", str(V),
                     "--padding_tgt", str(V), "--stride", "1", "--queue_size", "8", "--loader", "otf_utt", "--batch_first",
                     "--cmn", "--cmvn_stats", str(work / "cmvn.stats"), "--output_dim", str(V), "--num_workers", "1",
                     "--sample_rate", "16000", "--feat_config", str(work / "fbank.conf"), "--TU_limit", "15000",
                     "--gain_range", "50,10", "--speed_rate", "0.9,1.0,1.1", "--spec_augment", "--log_per_n_frames", "200",
                     "--max_len", "1600", "--lctx", "1", "--rctx", "1", "--model_lctx", "21", "--model_rctx", "21",
                     "--model_stride", "4", "--local-rank=0", "transducer", lst, str(out / ("train_%s_%d.WORKER-ID.log" % (mode, nb))),
                     str(ck)]
            for pair in filter(None, genv.get("COST_ARGS", "").split("|")):      # debugging: --flag:value|--flag:value
                k, _, v = pair.partition(":")
                if v == "OFF":
                    train.remove(k)
                else:
                    train[train.index(k) + 1] = v
            r0, t0 = resource.getrusage(resource.RUSAGE_CHILDREN), time.perf_counter()
            # (a failing reference script does not exit: its loader threads keep the process alive -- bound every run)
            r = subprocess.run(train, env=env, cwd=str(work), capture_output=True, text=True, timeout=420)
            wall = time.perf_counter() - t0
            r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
            cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
            assert r.returncode == 0, r.stderr[-6000:]
            stats = [l for l in r.stdout.splitlines() if l.startswith("train_graph")]
            if genv.get("PIKA_TRAIN_GRAPH_DEBUG") in ("2", "3"):
                print("\n".join(stats[:int(os.environ.get("COST_STAT_LINES", "60"))]))
            log = Path(str(out / ("train_%s_%d.0.log" % (mode, nb))))
            rows.append({"batches": nb, "wall_s": wall, "cpu_s": cpu, "train_graph": stats[-1:] if stats else None,
                         "log_tail": log.read_text().splitlines()[-int(os.environ.get("COST_LOG_LINES", "2")):]
                         if log.exists() else None})
            shutil.rmtree(ck, ignore_errors=True)
        if len(rows) < 2:
            print(mode, rows)
            continue
        d = rows[1]["batches"] - rows[0]["batches"]
        res[mode] = {"runs": rows, "ms_per_step_wall": (rows[1]["wall_s"] - rows[0]["wall_s"]) / d * 1e3,
                     "ms_per_step_process_cpu": (rows[1]["cpu_s"] - rows[0]["cpu_s"]) / d * 1e3}
        print(mode, json.dumps(res[mode], indent=1), flush=True)
    (out / "train_step_cost.json").write_text(json.dumps(res, indent=1))
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    if sys.argv[1] == "cost":
        train_cost(sys.argv[2])
    else:
        {"stage": stage, "clean": clean}.get(sys.argv[1], lambda: run(sys.argv[2]))()
