"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports
every symbol include/*.h declares; the ctypes table and the headers agree."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(pika_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_headers_declare_something():
    syms = declared_symbols()
    assert "pika_rnnt_loss_forward" in syms and "pika_rnnt_loss_backward" in syms


def test_library_builds_loads_and_exports_every_declared_symbol():
    from pika_amd import build
    path = build.build()
    handle = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(handle, name), "libpika_amd.so does not export %s" % name


def declared_prototypes():
    """name -> list of parameter declarations, parsed from include/*.h."""
    protos = {}
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        for name, args in re.findall(r"\b(pika_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
            args = " ".join(args.split())
            protos[name] = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
    return protos


def test_ctypes_table_matches_headers():
    from pika_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    protos = declared_prototypes()
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        decl = protos[name]
        assert len(decl) == len(argtypes), "%s: header has %d parameters, ctypes table %d" % (name, len(decl), len(argtypes))
        for d, t in zip(decl, argtypes):   # pointers <-> c_void_p, scalars <-> scalars
            assert ("*" in d) == (t is ctypes.c_void_p), "%s: parameter %r bound as %s" % (name, d, t)
    lib = _lib.lib()
    assert lib.pika_amd_abi_version() == _lib.ABI_VERSION


def test_workspace_size_formula():
    from pika_amd import _lib
    lib = _lib.lib()
    # four skewed f32 planes [B][D][W] (D = T+U1-1, W = 64 for U1 <= 64) + fp64 offsets
    # [2][B][D] + ll [2][B] + 16 bytes of row metadata per lattice cell
    def expect(B, T, U1, W):
        D = T + U1 - 1
        return 4 * B * D * W * 4 + (2 * B * D + 2 * B) * 8 + B * T * U1 * 16
    assert lib.pika_rnnt_workspace_bytes(32, 1000, 51) == expect(32, 1000, 51, 64)
    assert lib.pika_rnnt_workspace_bytes(1, 10, 65) == expect(1, 10, 65, 128)
    assert lib.pika_rnnt_workspace_bytes(3, 7, 200) == expect(3, 7, 200, 256)
    assert lib.pika_rnnt_workspace_bytes(0, 10, 5) == 0
    assert lib.pika_rnnt_workspace_bytes(1, 10, 1025) == 0


def test_bad_arguments_are_rejected_without_touching_the_gpu():
    from pika_amd import _lib
    lib = _lib.lib()
    # null pointers / bad dims are refused before any launch
    assert lib.pika_rnnt_loss_forward(None, None, None, None, 1, 1, 1, 1, 0, None, None, None) == -1
    assert lib.pika_rnnt_loss_forward(None, None, None, None, 0, 1, 1, 1, 0, None, None, None) == -1
    assert lib.pika_rnnt_loss_backward(None, None, None, 1, 1, 2000, 4, 0, None, None, None, None) == -2
    assert lib.pika_rnnt_loss_backward(None, None, None, 1, 1, 1, 4, 7, None, None, None, None) == -1


def test_product_op_refuses_cpu_tensors():
    import torch
    from pika_amd.rnnt import RNNTLoss
    lp = torch.zeros(1, 2, 2, 3)
    with pytest.raises(RuntimeError, match="HIP device"):
        RNNTLoss(blank=0, reduction="sum").apply(lp, torch.zeros(1, 1, dtype=torch.int32),
                                                 torch.ones(1, dtype=torch.int32),
                                                 torch.ones(1, dtype=torch.int32))


def test_dropin_import_surface():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "pika_amd", "dropin"))
    try:
        from warp_rnnt import RNNTLoss  # the reference's import line
        fn = RNNTLoss(blank=0, reduction="sum").apply
        assert callable(fn)
    finally:
        sys.path.pop(0)
