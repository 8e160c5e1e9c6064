"""The UNCHANGED reference training script run through `pika_amd.launch` (BASELINE.json configs[0],
CPU plumbing, world_size 1): argument plumbing, drop-in import surface, kaldi/torch._six shims,
loader file formats, model, BMUF (gloo), Logger, whole-module checkpoint.  The two GPU-only ops are
replaced by the oracles through a test-only preload (tests/cpu_plumbing.py).  Needs /root/reference,
which exists in the build container only."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SCRIPT = "/root/reference/trainer/train_transducer_bmuf_otfaug.py"
sys.path.insert(0, HERE)
from test_loader import make_corpus  # noqa: E402
from bmuf_common import free_port  # noqa: E402


def test_argv_and_shims():
    from pika_amd import launch
    assert launch.fix_argv(["--local-rank=3", "x"]) == ["--local_rank", "3", "x"]
    os.environ["LOCAL_RANK"] = "5"
    try:
        assert launch.fix_argv(["a"]) == ["a", "--local_rank", "5"]
        assert launch.fix_argv(["--local_rank", "1"]) == ["--local_rank", "1"]
    finally:
        del os.environ["LOCAL_RANK"]
    sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
    import editdistance
    assert editdistance.eval([1, 2, 3], [1, 3]) == 1 and editdistance.eval([], [4, 4]) == 2
    assert editdistance.eval("kitten", "sitting") == 3


@pytest.mark.skipif(not os.path.exists(SCRIPT), reason="reference tree not present on this box")
def test_unchanged_training_script_runs_one_epoch(tmp_path):
    lst, conf, pcms, labels = make_corpus(tmp_path, n_utts=4, seed=8, lo=14000, hi=20000)
    cmvn = tmp_path / "cmvn.stats"
    D = 80
    rng = np.random.default_rng(0)
    n, mean = 1000.0, rng.normal(8, 1, D)
    s1 = np.concatenate((mean * n, [n]))
    s2 = np.concatenate(((mean ** 2 + 4.0) * n, [0.0]))
    cmvn.write_text(" [\n  " + " ".join("%.10g" % v for v in s1) + "\n  " + " ".join("%.10g" % v for v in s2) + " ]\n")
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               PYTHONPATH=os.pathsep.join([ROOT, HERE]), OMP_NUM_THREADS="8")
    cmd = [sys.executable, "-m", "pika_amd.launch", "--preload", "cpu_plumbing", SCRIPT,
           "--verbose", "--optim", "sgd", "--initial_lr", "0.003", "--final_lr", "0.0001", "--grad_clip", "3.0",
           "--num_batches_per_epoch", "2", "--num_epochs", "1", "--momentum", "0.9", "--block_momentum", "0.9",
           "--sync_period", "1", "--feats_dim", "80", "--cuda", "--batch_size", "2", "--encoder_type", "transformer",
           "--enc_layers", "2", "--decoder_type", "transformer", "--dec_layers", "1", "--rnn_type", "LSTM",
           "--rnn_size", "64", "--embd_dim", "16", "--dropout", "0.0", "--padding_idx", "50", "--padding_tgt", "50",
           "--stride", "1", "--queue_size", "4", "--loader", "otf_utt", "--batch_first", "--cmn",
           "--cmvn_stats", str(cmvn), "--output_dim", "50", "--num_workers", "1", "--sample_rate", "16000",
           "--feat_config", conf, "--TU_limit", "15000", "--gain_range", "50,10", "--speed_rate", "0.9,1.0,1.1",
           "--log_per_n_frames", "1", "--max_len", "1600", "--lctx", "1", "--rctx", "1", "--model_lctx", "21",
           "--model_rctx", "21", "--model_stride", "4", "--local-rank=0",
           "transducer", lst, str(tmp_path / "train.WORKER-ID.log"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    log = (tmp_path / "train.0.log").read_text()
    assert "Training Finished" in log and "model proto: transducer" in log and "Loss:" in log
    losses = [float(l.split("Loss:")[1].split()[0]) for l in log.splitlines() if l.startswith("Loss:")]
    assert losses and all(np.isfinite(losses))
    sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
    model = torch.load(out / "model.epoch.0.0", weights_only=False)   # whole-module pickle (:363-366)
    assert type(model).__module__ == "pika_amd.model.transducer" and model.fc2.out_features == 50


LAS_SCRIPT = "/root/reference/trainer/train_las_bmuf_otfaug.py"


@pytest.mark.skipif(not os.path.exists(LAS_SCRIPT), reason="reference tree not present on this box")
def test_unchanged_las_training_script_runs_one_epoch(tmp_path):
    """the UNCHANGED LAS rescorer training script (train_las_bmuf_otfaug.py) through
    `pika_amd.launch`: otf loader with SOS/EOS targets -> shared (frozen) transducer encoder loaded from a
    whole-module pickle -> `model.las` Net -> its LASLossCompute -> Nesterov SGD -> BMUF sync -> checkpoint.
    (The script only runs with a shared encoder: without one `len_batch` is never assigned, :205-216.)"""
    from types import SimpleNamespace
    lst, conf, pcms, labels = make_corpus(tmp_path, n_utts=4, seed=9, lo=14000, hi=20000)
    sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
    from model.transducer import Net as TransducerNet
    torch.manual_seed(3)
    opt = SimpleNamespace(rnn_size=64, local_rank=0, decoder_type="transformer", brnn=False, encoder_type="transformer",
                          dropout=0.0, enc_layers=2, dec_layers=1, embd_dim=16, padding_idx=50)
    shared = tmp_path / "shared.mdl"
    torch.save(TransducerNet(opt, 240, 50), str(shared))
    out = tmp_path / "out"
    out.mkdir()
    cmvn = tmp_path / "cmvn.stats"
    rng = np.random.default_rng(0)
    n, mean = 1000.0, rng.normal(8, 1, 80)
    cmvn.write_text(" [\n  " + " ".join("%.10g" % v for v in np.concatenate((mean * n, [n]))) + "\n  " +
                    " ".join("%.10g" % v for v in np.concatenate(((mean ** 2 + 4.0) * n, [0.0]))) + " ]\n")
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               PYTHONPATH=os.pathsep.join([ROOT, HERE]), OMP_NUM_THREADS="8")
    # the options of egs/train_las_rescorer_bmuf_otfaug.sh at toy sizes
    cmd = [sys.executable, "-m", "pika_amd.launch", "--preload", "cpu_plumbing", LAS_SCRIPT,
           "--verbose", "--optim", "sgd", "--initial_lr", "0.003", "--final_lr", "0.0003", "--enc_loss_scale", "0.0",
           "--dec_loss_scale", "1.0", "--grad_clip", "3.0", "--lr", "0.001", "--cmn", "--cmvn_stats", str(cmvn),
           "--num_batches_per_epoch", "2", "--num_epochs", "1", "--momentum", "0.9", "--block_momentum", "0.9",
           "--sync_period", "1", "--feats_dim", "80", "--cuda", "--batch_size", "2",
           "--encoder_type", "rnn", "--decoder_type", "rnn", "--brnn", "--enc_layers", "2", "--dec_layers", "2",
           "--rnn_type", "LSTM", "--rnn_size", "32", "--embd_dim", "16", "--dropout", "0.2", "--global_attention", "mlp",
           "--input_dim", "64", "--output_dim", "50", "--padding_idx", "50", "--padding_tgt", "50", "--SOS", "0", "--EOS", "1",
           "--shared_encoder_model", str(shared), "--encoder_lctx", "21", "--encoder_rctx", "21", "--encoder_stride", "4",
           "--stride", "1", "--queue_size", "4", "--loader", "otf_utt", "--batch_first",
           "--num_workers", "1", "--sample_rate", "16000", "--feat_config", conf, "--TU_limit", "15000",
           "--gain_range", "50,10", "--speed_rate", "0.9,1.0,1.1", "--log_per_n_frames", "1", "--max_len", "1600",
           "--lctx", "1", "--rctx", "1", "--local-rank=0",
           "las", lst, str(tmp_path / "las.WORKER-ID.log"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    log = (tmp_path / "las.0.log").read_text()
    assert "Training Finished" in log and "model proto: las" in log and "DecLoss:" in log
    losses = [float(l.split("DecLoss:")[1].split()[0]) for l in log.splitlines() if "DecLoss:" in l]
    assert losses and all(np.isfinite(losses))
    model = torch.load(out / "model.epoch.0.0", weights_only=False)
    assert type(model).__module__ == "pika_amd.model.las" and model.dec_proj.out_features == 50


MBR_SCRIPT = "/root/reference/trainer/train_transducer_mbr_bmuf_otfaug.py"


@pytest.mark.skipif(not os.path.exists(MBR_SCRIPT), reason="reference tree not present on this box")
def test_unchanged_mbr_training_script_runs_one_epoch(tmp_path):
    """the UNCHANGED MBR fine-tuning script through `pika_amd.launch`: n-best generation with the
    drop-in TransducerDecoder (beam 3, beam_prune off) inside the training loop, its inline joint + RNN-T loss
    (a plain log-softmax tensor: the eager path of the loss), edit distances from the `editdistance` shim, the dense
    one-hot MBR gradient, BMUF sync, checkpoint."""
    lst, conf, pcms, labels = make_corpus(tmp_path, n_utts=4, seed=8, lo=14000, hi=20000)
    cmvn = tmp_path / "cmvn.stats"
    D = 80
    rng = np.random.default_rng(0)
    n, mean = 1000.0, rng.normal(8, 1, D)
    s1 = np.concatenate((mean * n, [n]))
    s2 = np.concatenate(((mean ** 2 + 4.0) * n, [0.0]))
    cmvn.write_text(" [\n  " + " ".join("%.10g" % v for v in s1) + "\n  " + " ".join("%.10g" % v for v in s2) + " ]\n")
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               PYTHONPATH=os.pathsep.join([ROOT, HERE]), OMP_NUM_THREADS="8")
    cmd = [sys.executable, "-m", "pika_amd.launch", "--preload", "cpu_plumbing", MBR_SCRIPT,
           "--verbose", "--optim", "sgd", "--initial_lr", "0.0003", "--final_lr", "0.0001", "--grad_clip", "3.0",
           "--num_batches_per_epoch", "2", "--num_epochs", "1", "--momentum", "0.9", "--block_momentum", "0.9",
           "--sync_period", "1", "--feats_dim", "80", "--cuda", "--batch_size", "2", "--encoder_type", "transformer",
           "--enc_layers", "2", "--decoder_type", "transformer", "--dec_layers", "1", "--rnn_type", "LSTM",
           "--rnn_size", "64", "--embd_dim", "16", "--dropout", "0.0", "--padding_idx", "50", "--padding_tgt", "50",
           "--stride", "1", "--queue_size", "4", "--loader", "otf_utt", "--batch_first", "--cmn",
           "--cmvn_stats", str(cmvn), "--output_dim", "50", "--num_workers", "1", "--sample_rate", "16000",
           "--feat_config", conf, "--TU_limit", "15000", "--gain_range", "50,10", "--speed_rate", "0.9,1.0,1.1",
           "--log_per_n_frames", "1", "--max_len", "1600", "--lctx", "1", "--rctx", "1", "--model_lctx", "21",
           "--model_rctx", "21", "--model_stride", "4", "--beam_size", "3", "--rnnt_scale", "0.01", "--local-rank=0",
           "transducer", lst, str(tmp_path / "mbr.WORKER-ID.log"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    log = (tmp_path / "mbr.0.log").read_text()
    assert "Training Finished" in log and "MBR Loss:" in log and "RNNT Loss:" in log
    vals = [float(l.split("MBR Loss:")[1].split()[0]) for l in log.splitlines() if l.startswith("MBR Loss:")]
    assert vals and all(np.isfinite(vals))
    sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
    model = torch.load(out / "model.epoch.0.0", weights_only=False)
    assert type(model).__module__ == "pika_amd.model.transducer"


DECODE_SCRIPT = "/root/reference/decoder/decode_transducer.py"


@pytest.mark.skipif(not os.path.exists(DECODE_SCRIPT), reason="reference tree not present on this box")
def test_unchanged_decode_script_and_the_eval_recipe_steps_behind_it(tmp_path):
    """The UNCHANGED decoding script (the command line of egs/eval_transducer.sh:74-100) through `pika_amd.launch`:
    Kaldi feature / label archives (utt loader), whole-module pickles of the transducer and of a LAS rescorer, an
    OpenFST binary n-gram LM (`kaldi.fstext.StdVectorFst.read` -> SortedMatcher shallow fusion), n-best list with
    scores and per-token LAS scores.  Like the reference, the script stops on the `None` its loader yields last
    (loader/utt_loader.py:237, decode_transducer.py:108): everything is written and flushed by then."""
    import struct
    from types import SimpleNamespace
    import fst_common as FC
    from pika_amd.loader import kaldi_io
    from pika_amd.decoder.ngram_fst import NgramFst
    V = 50
    rng = np.random.default_rng(4)
    utts = [("utt%d" % i, rng.normal(0, 1, (n, 80)).astype(np.float32)) for i, n in enumerate([130, 150, 121, 160])]
    kaldi_io.write_matrix_ark(str(tmp_path / "feats.ark"), utts)
    kaldi_io.write_int_vectors(str(tmp_path / "labels.ark"), [(k, np.array([1, 2, 3])) for k, _ in utts], binary=False)
    (tmp_path / "sym.map").write_text("".join("s%d %d\n" % (i, i) for i in range(V + 1)))
    sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
    from model.transducer import Net
    from model import las
    torch.manual_seed(3)
    opt = SimpleNamespace(rnn_size=64, local_rank=0, decoder_type="transformer", brnn=False, encoder_type="transformer",
                          dropout=0.0, enc_layers=2, dec_layers=1, embd_dim=16, padding_idx=V)
    torch.save(Net(opt, 240, V), str(tmp_path / "model.mdl"))
    lopt = SimpleNamespace(rnn_size=32, encoder_type="rnn", rnn_type="LSTM", brnn=True, enc_layers=1, dropout=0.0,
                           use_downsampler=False, embd_dim=12, num_heads=1, sampling_decoder=False, input_feed=1,
                           dec_layers=1, global_attention="mlp", coverage_attn=False, context_gate=None, copy_attn=False)
    torch.save(las.Net(lopt, 64, V + 1, V + 1), str(tmp_path / "las.mdl"))
    torch.manual_seed(4)
    torch.save(las.Net(lopt, 64, V + 1, V + 1), str(tmp_path / "las_bw.mdl"))
    n, arcs, finals, params = FC.bigram_arcs(V)
    ref = NgramFst.from_arcs(n, arcs, finals)
    with open(tmp_path / "g.fst", "wb") as f:          # OpenFST binary (vector / standard, no symbol tables)
        def s_(t): return struct.pack("<i", len(t)) + t.encode()
        f.write(struct.pack("<i", 2125659606) + s_("vector") + s_("standard") + struct.pack("<iiQqqq", 2, 0, 0, 0, n, len(arcs)))
        for st in range(n):
            lo, hi = ref.offsets[st], ref.offsets[st + 1]
            f.write(struct.pack("<fq", float(ref.final[st]), hi - lo))
            for j in range(lo, hi):
                f.write(struct.pack("<iifi", int(ref.ilabel[j]), int(ref.ilabel[j]), float(ref.weight[j]), int(ref.nextstate[j])))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, HERE]), OMP_NUM_THREADS="8")
    cmd = [sys.executable, "-m", "pika_amd.launch", "--preload", "cpu_plumbing", DECODE_SCRIPT,
           "--verbose", "--cuda", "--min_len", "50", "--blk", "0", "--batch_first", "--beam_size", "3", "--output_scores",
           "--sm_scale", "0.8", "--batch_size", "2", "--n_best", "2", "--SOS", "0", "--EOS", str(V), "--padding_idx", str(V),
           "--loader", "utt", "--lctx", "1", "--rctx", "1", "--feats_dim", "80", "--model_lctx", "21", "--model_rctx", "21",
           "--model_stride", "4", "--fst_lm", str(tmp_path / "g.fst"), "--fst_lm_scale", "0.3", "--nonblk_reward", "0.5",
           "--max_num_arcs", str(params["max_num_arcs"]), "--max_id", str(params["max_id"]),
           "--backoff_id", str(params["backoff_id"]), "--disambig_ids", ",".join(str(d) for d in params["disambig_ids"]),
           "--las_rescorer_model", str(tmp_path / "las.mdl"), "--las_rescorer_bw_model", str(tmp_path / "las_bw.mdl"),
           "--symbols_map", str(tmp_path / "sym.map"), str(tmp_path / "model.mdl"),
           "ark:" + str(tmp_path / "feats.ark"), "ark:" + str(tmp_path / "labels.ark"), str(tmp_path / "hyp.txt")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=900)
    assert "cannot unpack non-iterable NoneType" in r.stderr, r.stderr[-3000:]     # the reference's own ending
    lines = (tmp_path / "hyp.txt").read_text().splitlines()
    assert len(lines) == 4 * 2
    for ln in lines:
        fields = ln.split(" ")
        labels = [t for t in fields[0].split("s") if t]
        assert all(1 <= int(t) < V for t in labels)
        vals = [float(v) for v in fields[1:]]
        # beam score + forward and backward log P(token | prefix), each over the labels + EOS
        assert len(vals) == 1 + 2 * (len(labels) + 1) and all(np.isfinite(vals))
        assert all(v <= 1e-6 for v in vals[1:])
    # the rest of egs/eval_transducer.sh:100-127 on that file: the reference's own n-best reranker (pure Python, run
    # as it is) and the drop-in one pick the same hypotheses; keys are attached and the result is scored
    from pika_amd.eval import nbest_rerank, scoring
    ref_tool = "/root/reference/egs/local/nbest_rerank.py"
    r2 = subprocess.run([sys.executable, ref_tool, "--las_rescore", "--nbest", "2", str(tmp_path / "hyp.txt"),
                         str(tmp_path / "raw_ref.hyp")], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr[-2000:]
    nbest_rerank.main(["--las_rescore", "--nbest", "2", str(tmp_path / "hyp.txt"), str(tmp_path / "raw.hyp")])
    picked = (tmp_path / "raw.hyp").read_text()
    assert picked == (tmp_path / "raw_ref.hyp").read_text() and len(picked.splitlines()) == 4
    hyp_lines = scoring.attach_keys((tmp_path / "labels.ark").read_text().splitlines(), picked.splitlines())
    assert [l.split()[0] for l in hyp_lines] == [k for k, _ in utts]
    wer = scoring.compute_wer(["%s %s" % (k, " ".join(l.split()[1:])) for k, l in zip([k for k, _ in utts], hyp_lines)],
                              hyp_lines)
    assert wer["words"] > 0 and wer["ins"] == wer["del"] == wer["sub"] == 0 and wer["sent_errs"] == 0


@pytest.mark.skipif(not os.path.exists(SCRIPT), reason="reference tree not present on this box")
def test_unchanged_training_script_two_workers(tmp_path):
    """SURVEY 8(e) at the level of the unchanged script: TWO worker processes (what torch.distributed.launch starts,
    egs/train_transducer_bmuf_otfaug.sh) with WORKER-ID lists and logs, BMUF block sync every batch through the gloo
    backend (RCCL on a GPU node), loss reduction across workers; both write a checkpoint and, because BMUF ends the
    epoch with a sync, the two checkpoints hold identical parameters."""
    lists = []
    for rank in (0, 1):
        d = tmp_path / str(rank)
        d.mkdir()
        lst, conf, _, _ = make_corpus(d, n_utts=4, seed=20 + rank, lo=14000, hi=20000)
        lists.append(lst)
    cmvn = tmp_path / "cmvn.stats"
    rng = np.random.default_rng(0)
    n, mean = 1000.0, rng.normal(8, 1, 80)
    cmvn.write_text(" [\n  " + " ".join("%.10g" % v for v in np.concatenate((mean * n, [n]))) + "\n  " +
                    " ".join("%.10g" % v for v in np.concatenate(((mean ** 2 + 4.0) * n, [0.0]))) + " ]\n")
    out = tmp_path / "out"
    out.mkdir()
    port = str(free_port())
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   PYTHONPATH=os.pathsep.join([ROOT, HERE]), OMP_NUM_THREADS="4")
        cmd = [sys.executable, "-m", "pika_amd.launch", "--preload", "cpu_plumbing", SCRIPT,
               "--verbose", "--optim", "sgd", "--initial_lr", "0.003", "--final_lr", "0.0001", "--grad_clip", "3.0",
               "--num_batches_per_epoch", "2", "--num_epochs", "1", "--momentum", "0.9", "--block_momentum", "0.9",
               "--sync_period", "1", "--feats_dim", "80", "--cuda", "--batch_size", "2", "--encoder_type", "transformer",
               "--enc_layers", "2", "--decoder_type", "transformer", "--dec_layers", "1", "--rnn_type", "LSTM",
               "--rnn_size", "64", "--embd_dim", "16", "--dropout", "0.0", "--padding_idx", "50", "--padding_tgt", "50",
               "--stride", "1", "--queue_size", "4", "--loader", "otf_utt", "--batch_first", "--cmn",
               "--cmvn_stats", str(cmvn), "--output_dim", "50", "--num_workers", "1", "--sample_rate", "16000",
               "--feat_config", conf, "--TU_limit", "15000", "--gain_range", "50,10", "--speed_rate", "0.9,1.0,1.1",
               "--log_per_n_frames", "1", "--max_len", "1600", "--lctx", "1", "--rctx", "1", "--model_lctx", "21",
               "--model_rctx", "21", "--model_stride", "4", "--local-rank=%d" % rank,
               "transducer", str(tmp_path / "WORKER-ID" / "data.lst"), str(tmp_path / "train.WORKER-ID.log"), str(out)]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      cwd=str(tmp_path)))
    errs = []
    for p in procs:
        try:
            _, err = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        errs.append(err)
    assert [p.returncode for p in procs] == [0, 0], (errs[0][-1500:], errs[1][-1500:])
    sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
    models = [torch.load(out / ("model.epoch.0.%d" % r), weights_only=False) for r in (0, 1)]
    for r in (0, 1):
        assert "Training Finished" in (tmp_path / ("train.%d.log" % r)).read_text()
    for (ka, a), (kb, b) in zip(models[0].state_dict().items(), models[1].state_dict().items()):
        assert ka == kb
        if a.dtype.is_floating_point and "running_" not in ka:
            assert torch.equal(a, b), ka
