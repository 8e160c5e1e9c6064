"""`bench.py --gpus N` must START N ranks itself (the recipe launches its workers itself,
egs/train_transducer_bmuf_otfaug.sh:155-156) and report n_gpus = N.  Here: the launch / rendezvous / barrier /
max-over-ranks plumbing under gloo on CPU with a step that does nothing (PIKA_BENCH_DRYRUN=1: no product compute, no
oracle, marked `dry_run` in the line it prints).  The measured N-rank runs are the driver's (SCALE_rNN.json) and
profiles/r2_bench_n2*.json (two ranks on one MI355X, gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def _run(argv, extra_env=None):
    env = dict(os.environ, PIKA_BENCH_DRYRUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE json line, from rank 0 only
    return json.loads(lines[0])


def test_gpus_flag_spawns_that_many_ranks():
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["dry_run"] is True
    assert out["backend"] == "gloo"


def test_single_rank_default_and_world_mismatch_is_loud():
    out = _run(["--steps", "2", "--warmup", "0"])
    assert out["n_gpus"] == 1
    env = dict(os.environ, PIKA_BENCH_DRYRUN="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_rccl_selfcheck_runs_under_gloo_with_two_ranks():
    """tools/rccl_selfcheck.py (what `bench.py --gpus N` runs first when N > 1): environment check, checksummed all-reduce
    against its bounds -- here on CPU tensors over gloo, world_size 2."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                        os.path.join(ROOT, "tools", "rccl_selfcheck.py"), "--cpu"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["world"] == 2 and d["checksum_ok"] and d["all_reduce_ms"] > 0 and len(d["devices"]) == 2


def test_gpus_flag_plumbing_at_eight_ranks():
    """The driver's largest launch: `bench.py --gpus 8` starts eight ranks, they rendezvous, time between barriers and rank 0
    prints ONE line with n_gpus = 8 (dry run: no compute)."""
    out = _run(["--gpus", "8", "--steps", "2", "--warmup", "1"])
    assert out["n_gpus"] == 8 and out["dry_run"] is True and out["backend"] == "gloo"


import pytest  # noqa: E402


@pytest.mark.gpu
def test_eight_rank_line_on_one_gpu_carries_what_a_scaling_record_is_judged_by(hip_device):
    """The whole N-rank line at world_size 8 -- eight ranks on ONE GPU over gloo (no multi-GPU box here; RCCL's first run is
    the driver's): small batch / short utterances, M1 + the train-step leg only.  The line must carry the N = 1 sub-run of rank
    0, `train_step.vs_n1`, the BMUF exchange time next to its xGMI bounds, and the `scaling_summary` block documented in
    profiles/README.md ("the N = 8 line")."""
    env = dict(os.environ, PIKA_BENCH_DEVICE="0", PIKA_BENCH_BACKEND="gloo", PIKA_BENCH_WATCHDOG="900", PIKA_BENCH_PIN="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PIKA_BENCH_DRYRUN"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--batch", "2", "--frames", "300",
                        "--labels", "8", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-decode", "--no-mbr",
                        "--no-m1-variants"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["value"] > 0
    ts = d["train_step"]
    assert "error" not in ts, ts
    assert ts["n1_sub_run"]["value"] > 0 and ts["vs_n1"] > 0 and abs(ts["speedup_over_n1"] - 8 * ts["vs_n1"]) < 1e-6
    bm = ts["bmuf"]
    assert bm["syncs"] >= 1 and bm["all_reduce_ms"] > 0 and bm["bound_direct_ms"] > 0 and bm["bound_ring_ms"] > bm["bound_direct_ms"]
    assert len(ts["per_rank"]) == 8
    assert "lstm_prediction_net" not in ts and "bf16x3" not in ts          # N = 1 legs only
    sm = d["scaling_summary"]
    assert sm["train_step_vs_n1"] == ts["vs_n1"] and sm["bmuf_all_reduce_ms"] == bm["all_reduce_ms"]
    assert "independent kernels" in sm["headline"] and sm["rccl"]["backend"] == "gloo"
