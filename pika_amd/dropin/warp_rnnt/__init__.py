"""Drop-in for `from warp_rnnt import RNNTLoss`
(/root/reference/trainer/train_transducer_bmuf_otfaug.py:25) backed by libpika_amd.so."""
from pika_amd.rnnt import RNNTLoss, rnnt_loss  # noqa: F401
