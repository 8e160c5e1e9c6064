"""pika_amd -- MI355X-native (gfx950) RNN-Transducer hot path behind PIKA's Python API.

Product code only: hand-written HIP kernels + the C ABI (``csrc/``, ``include/*.h``) and the
host-side mirror of the reference's operator interface (``warp_rnnt.RNNTLoss`` ...).  The
drop-in import surface (``warp_rnnt``, ``trainer.*``, ``decoder.*``, ``loader.*``, ``utils.*``)
lives under ``pika_amd/dropin`` and is put on ``sys.path`` by ``pika_amd.launch``.

Nothing in this package imports ``oracle/`` and there is no CPU fallback: ops raise if the
HIP library is missing or if they are handed non-GPU tensors.
"""
__version__ = "0.1.0"
