"""Debug: which variation of the small graphed train step survives hipStreamEndCapture.  Each case in its own process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
    "ragged_w2": {"PROBE_RAGGED": "7", "PROBE_WARMUP": "2"},
    "ragged_ref_first": {"PROBE_RAGGED": "7", "PROBE_REF": "1"},
    "ragged_script_loop": {"PROBE_RAGGED": "7", "PROBE_SCRIPT": "1"},
    "ragged_script_loop_8": {"PROBE_RAGGED": "7", "PROBE_SCRIPT": "1", "PROBE_N": "8"},
    "ragged_script_loop_8_rebuild": {"PROBE_RAGGED": "7", "PROBE_SCRIPT": "1", "PROBE_N": "8", "PROBE_REBUILD": "3"},
}


def child():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
    import torch
    import test_train_step_gpu as T
    from pika_amd import gemm as G
    from pika_amd.train_graph import GraphedTrainStep
    V = int(os.environ.get("PROBE_V", "512"))
    model, loss_fn, _, fused_optim = T._small_step_harness("cuda:0", 0.0, V=V)
    g = torch.Generator().manual_seed(21)
    rag = os.environ.get("PROBE_RAGGED")
    batches = [T._batch("cuda:0", g, 4, int(os.environ.get("PROBE_T", "300")), int(os.environ.get("PROBE_U", "11")), V,
                        pad_from=int(rag) if rag else None) for _ in range(4)]
    if os.environ.get("PROBE_SETENV"):
        os.environ["PIKA_TRAIN_GRAPH"] = "0"
        os.environ["PIKA_TRAIN_GRAPH"] = "1"
    G.PRECISION = "mixed"
    fused_optim.install()
    n = int(os.environ.get("PROBE_N", "4"))
    while len(batches) < n:
        batches.append(T._batch("cuda:0", g, 4, 300, 11, V, pad_from=int(rag) if rag else None))
    if os.environ.get("PROBE_REF"):
        import copy
        ref = copy.deepcopy(model)
        print("ref", T._script_loop(ref, batches, rebuild_every=3)[-1], flush=True)
    if os.environ.get("PROBE_SCRIPT"):
        from pika_amd import train_graph
        train_graph.AUTO = True
        print("script", T._script_loop(model, batches, rebuild_every=int(os.environ.get("PROBE_REBUILD", "0"))), flush=True)
        print(model._step_graphs.stats, flush=True)
        return
    gs = GraphedTrainStep(model, loss_fn, lambda: torch.optim.SGD(model.parameters(), 1e-4, momentum=0.9, nesterov=True),
                          clip=3.0, warmup=int(os.environ.get('PROBE_WARMUP', '1')))
    for b in batches:
        print("loss", float(gs(*b)), gs.state.stats, gs.state.broken, flush=True)
    print("kinds", [e.kind for e in gs.graphs.values()], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for name, env in CASES.items():
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env),
                               capture_output=True, text=True)
            print("==", name, "rc", r.returncode)
            print("\n".join(r.stdout.strip().splitlines()[-5:]))
            if r.returncode:
                print("\n".join(l for l in r.stderr.strip().splitlines() if "File \"/root/repo" in l or "Error" in l or "fault" in l)[-1500:])
