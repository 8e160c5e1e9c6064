"""oracle/pika_ref.py -- TEST INFRASTRUCTURE: imports the REFERENCE's own Python modules from
/root/reference (read-only) under the compatibility shims of SURVEY.md 8c, to generate golden
vectors in THIS container.  /root/reference does not exist on the GPU box, so nothing at GPU
test time imports this file's `load_reference()`; the committed fixtures travel instead.

Shims (all applied outside the reference tree):
  1. torch.Tensor.cuda -> identity                    (transducer.py:91 hard-codes SOS.cuda())
  2. integer-tensor `/` -> floor division             (beam_transducer.py:125, torch<=1.4 idiom)
  3. torch.cuda.LongTensor/FloatTensor -> CPU types   (transducer_decoder.py:166)
"""
import importlib
import sys

import torch

import os


def _reference_root():
    """/root/reference in the build container; on the GPU box (where it does not exist) the staged copy of the hot-path
    modules that tools/stage_reference.py puts under the git-ignored _ref_scratch/ -- it travels with the snapshot like a
    built .so -- so that bench.py's cpu_baseline legs can time the REFERENCE itself on that host.  PIKA_REF_ROOT overrides."""
    env = os.environ.get("PIKA_REF_ROOT")
    if env:
        return env
    if os.path.isdir("/root/reference"):
        return "/root/reference"
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_ref_scratch", "reference")


REF = _reference_root()


def available():
    return os.path.isfile(os.path.join(REF, "trainer", "model", "transducer.py"))


def seeded_state_dict(module, seed, scale=0.1):
    """Deterministic weights for ANY module with a given key/shape layout: tensors are filled in
    state_dict key order from one seeded generator.  Used on the reference model when a golden
    file is made and on our model at test time, so both carry identical weights without
    shipping them."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in module.state_dict().items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            sd[k] = torch.rand(v.shape, generator=g) + 0.5
        elif v.dtype.is_floating_point:
            t = torch.randn(v.shape, generator=g) * scale
            if k.endswith("layer_norm.weight") or (k.endswith(".weight") and v.dim() == 1):
                t = t + 1.0  # norm gains around 1
            sd[k] = t
        else:
            sd[k] = v.clone()  # integer buffers (attention mask)
    return sd


def apply_shims():
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.LongTensor = torch.LongTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    _true = torch.Tensor.__truediv__

    def _legacy_div(a, b):
        if isinstance(a, torch.Tensor) and not a.dtype.is_floating_point and not a.dtype.is_complex \
                and (not isinstance(b, torch.Tensor) or not b.dtype.is_floating_point) \
                and not isinstance(b, float):
            return torch.div(a, b, rounding_mode="floor")
        return _true(a, b)
    torch.Tensor.__truediv__ = _legacy_div


def load_reference(*names):
    """Import reference modules by dotted name, e.g. 'trainer.model.transducer'."""
    apply_shims()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k.split(".")[0] in ("trainer", "decoder", "utils", "loader")]:
        if not getattr(sys.modules[k], "__file__", None) or REF not in (sys.modules[k].__file__ or ""):
            del sys.modules[k]  # drop our drop-in modules of the same name
    mods = [importlib.import_module(n) for n in names]
    return mods[0] if len(mods) == 1 else mods
