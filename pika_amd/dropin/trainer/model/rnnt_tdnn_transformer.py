"""Drop-in for trainer.model.rnnt_tdnn_transformer."""
from pika_amd.model.encoder import Net  # noqa: F401
