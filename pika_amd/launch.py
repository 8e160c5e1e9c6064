"""Runs an UNCHANGED PIKA script against the drop-in packages:

    python -m pika_amd.launch [--preload mod ...] [--legacy-int-div] /path/to/pika/trainer/train_transducer_bmuf_otfaug.py <its args>

What it absorbs (SURVEY.md 8b "legacy-runtime conventions"), without touching the script:
  * import resolution: `pika_amd/dropin` goes first on sys.path, so `warp_rnnt`, `trainer.*`,
    `model.*`, `decoder.*`, `loader.*`, `utils.*`, `kaldi` (matrix / util / fstext), `editdistance` resolve to this
    repository even though the script's own directory is sys.path[0] under plain `python script`;
  * `from torch._six import inf` (removed in torch >= 2.0);
  * `--local-rank=N` (what torch.distributed.launch passes today) -> `--local_rank N`, or the
    LOCAL_RANK environment variable when neither is given;
  * `torch.load` of whole-module pickles (`weights_only` now defaults to True);
  * the per-batch `clip_grad_norm_(.., norm_type=inf)` + `optim.SGD(.., nesterov=True).step()` of the training
    scripts run as three multi-tensor HIP launches (pika_amd/optim.py; stock torch for everything else);
  * `model.forward(...)` / `loss.backward()` of a training step on a HIP device replay two hipGraphs captured per batch
    shape (pika_amd/train_graph.py): ~600 launches of host work per step become two;
  * integer-tensor `/` as integer division (torch <= 1.4 semantics) ONLY on request (`--legacy-int-div`): the one
    reference line that relies on it (decoder/beam_transducer.py:125) lives in a module the drop-in `decoder`
    package replaces, so the process-wide patch is off unless a user script of its own needs it.
"""
import importlib
import math
import os
import runpy
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
DROPIN = os.path.join(HERE, "dropin")


def install_shims(legacy_int_div=False):
    import pika_amd  # noqa: F401  (first: turns the HIP runtime's graph packet-capture fast path off before torch touches HIP)
    import torch
    if "torch._six" not in sys.modules:
        six = types.ModuleType("torch._six")
        six.inf = math.inf
        six.string_classes = (str,)
        sys.modules["torch._six"] = six
        torch._six = six
    real_load = torch.load

    def load(*a, **kw):
        kw.setdefault("weights_only", False)
        return real_load(*a, **kw)
    torch.load = load
    # clip_grad_norm_(..., norm_type=inf) + Nesterov optim.SGD.step() of the training scripts as three HIP launches
    # (pika_amd/optim.py); every other use falls through to torch
    from . import optim as _optim
    _optim.install()
    # the model's forward / backward of a TRAINING step as two hipGraph replays behind Net.forward (pika_amd/train_graph.py;
    # PIKA_TRAIN_GRAPH=0: the eager launch sequence)
    from . import train_graph as _tg
    _tg.AUTO = True
    if legacy_int_div:
        true_div = torch.Tensor.__truediv__

        def legacy_div(a, b):
            if not a.dtype.is_floating_point and not a.dtype.is_complex and (
                    isinstance(b, int) or (isinstance(b, torch.Tensor) and not b.dtype.is_floating_point
                                           and not b.dtype.is_complex)):
                return torch.div(a, b, rounding_mode="trunc")     # C semantics, as torch <= 1.4
            return true_div(a, b)
        torch.Tensor.__truediv__ = legacy_div


def fix_argv(argv):
    out, seen = [], False
    for a in argv:
        if a.startswith("--local-rank"):
            a = a.replace("--local-rank", "--local_rank", 1)
        if a.startswith("--local_rank"):
            seen = True
        if a.startswith("--local_rank="):
            out.extend(a.split("=", 1))
            continue
        out.append(a)
    if not seen and "LOCAL_RANK" in os.environ:
        out.extend(["--local_rank", os.environ["LOCAL_RANK"]])
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    preload, legacy_int_div = [], False
    while argv and argv[0] in ("--preload", "--legacy-int-div"):
        if argv[0] == "--legacy-int-div":
            legacy_int_div, argv = True, argv[1:]
            continue
        preload.append(argv[1])
        argv = argv[2:]
    if not argv:
        raise SystemExit(__doc__)
    script = argv[0]
    root = os.path.dirname(HERE)
    for p in (root, DROPIN):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    if os.environ.get("PIKA_LAUNCH_WATCHDOG"):      # diagnostics on a GPU box: dump every thread's stack and exit after N s
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PIKA_LAUNCH_WATCHDOG"]), exit=True)
    install_shims(legacy_int_div)
    for m in preload:
        importlib.import_module(m)
    sys.argv = [script] + fix_argv(argv[1:])
    # run_path would put the script's directory at sys.path[0]; keep ours in front instead
    code_dir = os.path.dirname(os.path.abspath(script))
    if code_dir not in sys.path:
        sys.path.append(code_dir)
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
