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
