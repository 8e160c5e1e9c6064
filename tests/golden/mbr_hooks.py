"""TEST INFRASTRUCTURE for the MBR row: runs the UNCHANGED reference script
trainer/train_transducer_mbr_bmuf_otfaug.py for one batch and dumps every parameter gradient right before its first
`optimizer.step()` (i.e. after the RNN-T backward :159 and the risk backward :235 of the script's own inline code).

    python tests/golden/mbr_hooks.py reference <out.npz>     # the reference's own trainer.model.* / decoder.* / bmuf
    python tests/golden/mbr_hooks.py dropin <out.npz>        # this repository's drop-in packages (CPU tensors)

Both modes: the same seeded weights (`--init_model`, oracle.pika_ref.seeded_state_dict), the same fixture loader
(tests/golden/mbr_fixture), the same CPU stand-in for the GPU-only loss (the fp32 oracle), `.cuda()` as identity, gloo
with one worker.  Nothing here is imported by the product."""
import os
import runpy
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
REF = os.environ.get("PIKA_REF_ROOT") or ("/root/reference" if os.path.isdir("/root/reference") else os.path.join(
    os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "_ref_scratch", "reference"))
SCRIPT = os.path.join(REF, "trainer", "train_transducer_mbr_bmuf_otfaug.py")
V = 40
MODEL_ARGS = ["--encoder_type", "transformer", "--enc_layers", "2", "--decoder_type", "transformer", "--dec_layers", "1",
              "--rnn_type", "LSTM", "--rnn_size", "64", "--embd_dim", "16", "--dropout", "0.0",
              "--padding_idx", str(V), "--output_dim", str(V)]


def script_args(init_model, log, outdir):
    return ["--optim", "sgd", "--initial_lr", "0.001", "--final_lr", "0.001", "--num_batches_per_epoch", "1",
            "--num_epochs", "1", "--momentum", "0.9", "--block_momentum", "0.9", "--sync_period", "5", "--cuda",
            "--loader", "otf_utt", "--beam_size", "3", "--rnnt_scale", "0.1", "--sm_scale", "0.9", "--blk", "0",
            "--model_lctx", "12", "--model_rctx", "12", "--model_stride", "4", "--local_rank", "0",
            "--init_model", init_model] + MODEL_ARGS + ["transducer", "unused.lst", log, outdir]


def compact(grads, n_samples=512):
    """Per parameter: up to 512 evenly strided gradient entries + [L2 norm, signed sum, max |g|, numel] (the
    prediction net alone has 3 M parameters -- too much for a committed fixture)."""
    import numpy as np
    out = {}
    for k, g in grads.items():
        f = np.asarray(g, np.float64).reshape(-1)
        stride = max(1, f.size // n_samples)
        out["s" + k[1:]] = f[::stride][:n_samples].astype(np.float32)
        out["m" + k[1:]] = np.array([np.sqrt((f * f).sum()), f.sum(), np.abs(f).max(), f.size])
    return out


def compare(got, want, rel=1e-3):
    """Asserts two `compact` dumps agree: samples to rel * max|g| (+1e-7), norms and sums to rel."""
    import numpy as np
    assert int(got["n"]) == int(want["n"])
    for i in range(int(want["n"])):
        k = "%03d" % i
        ms, mg = want["m" + k], got["m" + k]
        assert ms[3] == mg[3], (i, ms[3], mg[3])
        tol = rel * ms[2] + 1e-7
        assert np.abs(got["s" + k] - want["s" + k]).max() <= tol, (i, np.abs(got["s" + k] - want["s" + k]).max(), tol)
        assert abs(mg[0] - ms[0]) <= rel * ms[0] + 1e-7, (i, "l2", mg[0], ms[0])
        assert abs(mg[1] - ms[1]) <= rel * ms[2] * np.sqrt(ms[3]) + 1e-6, (i, "sum", mg[1], ms[1])


def install_dump_hook(out_path):
    import importlib
    import numpy as np
    import torch
    seen = {}
    D = importlib.import_module("decoder.transducer_decoder")
    real_decode = D.TransducerDecoder.decode_batch

    def decode_batch(self, *a, **k):       # keep the N-best the script's own decode produced (:112-117)
        ret, enc = real_decode(self, *a, **k)
        seen["hyps"] = [[[int(e) for e in h] for h in row] for row in ret["predictions"]]
        seen["scores"] = [[float(v) for v in row] for row in ret["scores"]]
        return ret, enc
    D.TransducerDecoder.decode_batch = decode_batch

    def step(self, closure=None):
        grads = {}
        i = 0
        for group in self.param_groups:
            for p in group["params"]:
                grads["g%03d" % i] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().numpy()
                i += 1
        L = max(len(h) for row in seen["hyps"] for h in row)
        hyps = np.full((len(seen["hyps"]), len(seen["hyps"][0]), L), -1, np.int64)
        for b, row in enumerate(seen["hyps"]):
            for j, h in enumerate(row):
                hyps[b, j, :len(h)] = h
        np.savez(out_path, n=np.array(i), hyps=hyps, scores=np.array(seen["scores"], np.float64), **compact(grads))
        raise SystemExit(0)
    torch.optim.SGD.step = step


def fake_cuda():
    import torch
    # the encoder's transformer layers hard-code dropout 0.2 (rnnt_tdnn_transformer.py:62-65): random masks are not
    # comparable across implementations, so both runs are made with dropout as the identity
    torch.nn.functional.dropout = lambda x, p=0.5, training=True, inplace=False: x
    torch.nn.Dropout.forward = lambda self, x: x
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.manual_seed = lambda *a, **k: None


def oracle_loss_module():
    """`warp_rnnt` for the reference-mode run: the fp32 CPU oracle behind the call shape of the script (:64,157)."""
    import numpy as np
    import torch
    from oracle import rnnt as O

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, lp, labels, tl, ul):
            costs, grads = O.rnnt_loss(lp.detach().numpy(), labels.numpy(), tl.numpy(), ul.numpy(), dtype=np.float32)
            ctx.save_for_backward(torch.from_numpy(grads))
            return torch.from_numpy(costs)

        @staticmethod
        def backward(ctx, g):
            (grads,) = ctx.saved_tensors
            return grads * g.view(-1, 1, 1, 1), None, None, None

    class RNNTLoss(object):
        def __init__(self, blank=0, reduction="sum"):
            pass

        def apply(self, lp, labels, tl, ul):
            return _Fn.apply(lp, labels, tl.int(), ul.int())
    m = types.ModuleType("warp_rnnt")
    m.RNNTLoss = RNNTLoss
    return m


def save_seeded_model(path):
    """A whole-module pickle of `model.transducer.Net` (whichever package `model` resolves to in this process) with
    the seeded weights both modes share.  As in tests/model_common.py the 1024-wide / 9-layer encoder hard-coded in
    transducer.Net is swapped for a 64-wide / 6-layer instance of the SAME encoder class (a random 1024-wide net
    amplifies fp32 rounding differences between two correct implementations by 1e3 through its 12 layers, which
    would make a 1e-3 comparison meaningless); the script takes the module as it is (`--init_model`)."""
    import argparse
    import importlib
    import torch
    from oracle.pika_ref import seeded_state_dict
    ap = argparse.ArgumentParser()
    for k in ("--encoder_type", "--decoder_type", "--rnn_type"):
        ap.add_argument(k)
    for k in ("--enc_layers", "--dec_layers", "--rnn_size", "--embd_dim", "--padding_idx", "--output_dim"):
        ap.add_argument(k, type=int)
    ap.add_argument("--dropout", type=float)
    opt = ap.parse_args(MODEL_ARGS)
    opt.local_rank, opt.brnn = 0, False
    Net = importlib.import_module("model.transducer").Net
    net = Net(opt, 240, V)
    net.encoder = type(net.encoder)(240, 0, opt.rnn_size, tdnn_nhid=64, tdnn_layers=6)
    net.pack_seq = False
    sd = seeded_state_dict(net, 1234)
    sd["fc2.bias"][0] += 1.5           # some blanks in the N-best: trajectories that advance in t AND u
    net.load_state_dict(sd)
    torch.save(net, path)


def main(mode, out_path):
    import tempfile
    import torch
    tmp = tempfile.mkdtemp(prefix="mbr_%s_" % mode)
    os.environ.update(WORLD_SIZE="1", RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=os.environ.get("MASTER_PORT", "29731"))
    fixture = os.path.join(HERE, "mbr_fixture")
    dropin = os.path.join(ROOT, "pika_amd", "dropin")
    if mode == "reference":
        # the reference's own trainer.model.*, decoder.*, trainer.bmuf, utils.*; third-party imports it cannot satisfy
        # here (warp_rnnt, editdistance, kaldi: all outside the code under test) are stood in for
        sys.path[:0] = [fixture, ROOT, os.path.join(REF, "trainer"), REF]
        from oracle.pika_ref import apply_shims
        apply_shims()
        six = types.ModuleType("torch._six")
        six.inf = float("inf")
        sys.modules["torch._six"] = six
        sys.modules["warp_rnnt"] = oracle_loss_module()
        import importlib.util
        for name in ("editdistance", "kaldi"):       # pure-Python stand-ins that ship with the drop-in tree
            spec = importlib.util.spec_from_file_location(
                name, os.path.join(dropin, name, "__init__.py"), submodule_search_locations=[os.path.join(dropin, name)])
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
        real_load = torch.load
        torch.load = lambda *a, **k: real_load(*a, **dict(k, weights_only=False))
        real_init = torch.distributed.init_process_group
        torch.distributed.init_process_group = lambda backend=None, **k: real_init(backend="gloo", **k)
    else:
        sys.path[:0] = [fixture, ROOT, dropin, TESTS]
        from pika_amd import launch
        launch.install_shims()
        import cpu_plumbing  # noqa: F401  (oracle loss + .cuda identity for the drop-in packages on CPU)
    fake_cuda()
    init = os.path.join(tmp, "init.mdl")
    save_seeded_model(init)
    install_dump_hook(out_path)
    torch.set_num_threads(4)
    sys.argv = [SCRIPT] + script_args(init, os.path.join(tmp, "train.log"), tmp)
    if os.path.join(REF, "trainer") not in sys.path:
        sys.path.append(os.path.join(REF, "trainer"))
    runpy.run_path(SCRIPT, run_name="__main__")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
