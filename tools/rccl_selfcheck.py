#!/usr/bin/env python
"""First-contact check of the N > 1 path (`bench.py --gpus N` runs it before anything else; also
`python -m torch.distributed.run --nproc-per-node N tools/rccl_selfcheck.py`).

The RCCL path of this repository had only ever run under gloo when this was written (one-GPU leases): a first run on
an 8-GPU node should produce a DIAGNOSIS, not a hang.  What is checked, each with its own failure text:

  1. environment: HSA_ENABLE_IPC_MODE_LEGACY=0 (the host driver only supports dmabuf IPC: without it RCCL fails with
     `hipIpcGetMemHandle: invalid argument`), one distinct GPU per rank (two ranks on one GPU deadlock RCCL's kernels);
  2. ONE all-reduce of the BMUF vector's size (90.3 M fp32 = 361 MB: what trainer/bmuf.py:76-100 exchanges every
     sync_period steps): checksum against the closed form, HIP-event time against the xGMI bounds of SURVEY 5.8
     (direct: 2 x bytes/N per link at 153 GB/s; ring: 2 (N-1)/N x bytes through one link);
  3. a collective BEHIND a hipGraph replay on the same stream (the train step replays graphs, then BMUF all-reduces).

Every step runs under a watchdog that dumps all stacks and exits instead of hanging the node.  Returns a dict for the
bench line (`rccl_selfcheck`); raises with the diagnosis on failure.  Works on CPU tensors / gloo for the tests."""
import faulthandler
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

LINK_GBPS = 153.0       # one xGMI link, per direction


def _fail(msg):
    raise RuntimeError("rccl_selfcheck: " + msg)


def selfcheck(dev, n_elems=90_300_000, watchdog_s=180):
    rank, world = dist.get_rank(), dist.get_world_size()
    on_gpu = dev.type == "cuda"
    faulthandler.dump_traceback_later(watchdog_s, exit=True)       # a hang becomes a stack dump + exit
    try:
        out = {"world": world, "backend": dist.get_backend(), "elements": n_elems}
        # -- 1. environment -------------------------------------------------------------------------------------------
        if on_gpu and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
            _fail("HSA_ENABLE_IPC_MODE_LEGACY is %r; export HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC) before launching "
                  "the ranks, or RCCL's first collective fails with hipIpcGetMemHandle: invalid argument"
                  % os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
        ident = "%s/%s" % (socket.gethostname(), (str(getattr(torch.cuda.get_device_properties(dev), "uuid", "")) or
                                                  torch.cuda.get_device_properties(dev).name + ":%d" % dev.index) if on_gpu
                           else "cpu:%d" % rank)
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        out["devices"] = idents
        if on_gpu and len(set(idents)) != world and os.environ.get("PIKA_BENCH_DEVICE") is None:
            _fail("%d ranks share GPUs (%s): launch one rank per GPU (LOCAL_RANK -> torch.cuda.set_device); two ranks on "
                  "one device deadlock inside RCCL's kernels" % (world, idents))
        # -- 2. the BMUF-sized all-reduce -----------------------------------------------------------------------------
        x = torch.full((n_elems,), float(rank + 1), dtype=torch.float32, device=dev)
        dist.all_reduce(x)                                  # connection set-up + first launch: untimed
        want = world * (world + 1) / 2.0
        x.fill_(float(rank + 1))
        if on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
        t0 = time.perf_counter()
        dist.all_reduce(x)
        if on_gpu:
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
        else:
            ms = (time.perf_counter() - t0) * 1e3
        probe = x[:: max(1, n_elems // 1024)]
        if not bool((probe == want).all()):
            _fail("all_reduce(SUM) of rank+1 over %d ranks gave %r, expected %r everywhere: the collective ran on the wrong "
                  "data (stream ordering?) or a rank did not take part" % (world, probe[:4].tolist(), want))
        nbytes = 4.0 * n_elems
        out.update(all_reduce_ms=ms, checksum_ok=True,
                   bound_direct_ms=2.0 * (nbytes / world) / (LINK_GBPS * 1e9) * 1e3,
                   bound_ring_ms=2.0 * (world - 1) / world * nbytes / (LINK_GBPS * 1e9) * 1e3)
        if on_gpu and ms > 20 * out["bound_ring_ms"] + 50:
            print("rccl_selfcheck: WARNING all-reduce of %.0f MB took %.1f ms, ring bound %.1f ms: RCCL is probably not on "
                  "xGMI (NCCL_DEBUG=INFO shows the transport; PCIe / sockets are 10-50x slower)"
                  % (nbytes / 1e6, ms, out["bound_ring_ms"]), file=sys.stderr, flush=True)
        # -- 3. a collective behind a graph replay --------------------------------------------------------------------
        if on_gpu:
            a = torch.ones(1 << 20, device=dev)
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                a.mul_(1.0)                                 # warm the kernel outside the capture
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                a.mul_(2.0)
            a.fill_(float(rank + 1))
            g.replay()                                      # a = 2 (rank + 1)
            dist.all_reduce(a)                              # must see the replay's result
            torch.cuda.synchronize()
            if abs(float(a[12345]) - 2.0 * want) > 1e-3:
                _fail("an all_reduce enqueued behind a hipGraph replay did not see the replay's result (%r, expected %r): "
                      "RCCL's stream is not ordered behind torch's current stream" % (float(a[12345]), 2.0 * want))
            out["graph_then_collective_ok"] = True
        return out
    finally:
        faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    cpu = "--cpu" in sys.argv or not torch.cuda.is_available()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not cpu:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend="gloo" if cpu else "nccl", init_method="env://", rank=rank, world_size=world)
    res = selfcheck(torch.device("cpu") if cpu else torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))),
                    n_elems=1_000_000 if cpu else 90_300_000)
    if rank == 0:
        import json
        print(json.dumps(res))
    dist.destroy_process_group()
