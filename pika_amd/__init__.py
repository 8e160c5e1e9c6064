"""pika_amd -- MI355X-native (gfx950) RNN-Transducer hot path behind PIKA's Python API.

Product code only: hand-written HIP kernels + the C ABI (``csrc/``, ``include/*.h``) and the
host-side mirror of the reference's operator interface (``warp_rnnt.RNNTLoss`` ...).  The
drop-in import surface (``warp_rnnt``, ``trainer.*``, ``decoder.*``, ``loader.*``, ``utils.*``)
lives under ``pika_amd/dropin`` and is put on ``sys.path`` by ``pika_amd.launch``.

Nothing in this package imports ``oracle/`` and there is no CPU fallback: ops raise if the
HIP library is missing or if they are handed non-GPU tensors.
"""
__version__ = "0.1.0"

import os as _os
import sys as _sys


def _hip_graph_workaround():
    """ROCm 7.0 HIP runtime, MI355X: with the runtime's "graph packet capture" fast path (AQL packets pre-built at
    hipGraphInstantiate; DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default) two graph execs replayed ALTERNATELY with ordinary
    launches in between -- the forward / backward graphs of pika_amd.train_graph around the eager loss, clip and SGD
    launches -- go wrong from about the tenth replay: a replay's kernels run on stale state (the training loss of a
    replayed loop leaves the eager loop's, bit-identical for nine steps, at the tenth; any host synchronisation placed
    between the launches moves the step at which it happens; with the fast path off the two loops agree step for step,
    tools/graph_dropout_diff.py).  The flag is read once, when the HIP runtime initialises, so it is set here, at import
    -- before torch makes its first HIP call in the normal order of imports (`pika_amd.launch`, bench.py and tests/conftest.py
    import this package first; a script that has already called torch.cuda.is_available() gets the eager step).  Decode (one exec replayed back to back) is
    not affected either way and costs the same with the flag off (171 vs 173 ms per batch at configs[4])."""
    if "DEBUG_CLR_GRAPH_PACKET_CAPTURE" in _os.environ:
        return _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "0"
    # Has the HIP runtime already read its flags?  torch.cuda.is_initialized() only knows about torch's own lazy init --
    # torch.cuda.is_available() / device_count() call hipGetDeviceCount without setting it -- so the question is put to the
    # process itself: the runtime opens /dev/kfd when it starts.  Unknown (no /proc) counts as "already started" when torch
    # is imported: the training step then stays eager, which is slower, never wrong.
    started = None
    try:
        started = False
        for fd in _os.listdir("/proc/self/fd"):
            try:
                if _os.readlink("/proc/self/fd/" + fd) == "/dev/kfd":
                    started = True
                    break
            except OSError:
                pass
    except OSError:
        started = None
    late = started if started is not None else ("torch" in _sys.modules)
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    return not late


# False: the HIP runtime was initialised before this package could turn the fast path off (or the user turned it on):
# pika_amd.train_graph then keeps the training step an eager launch sequence
HIP_GRAPHS_SAFE_TO_ALTERNATE = _hip_graph_workaround()
