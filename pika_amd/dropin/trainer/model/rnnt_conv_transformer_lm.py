"""Drop-in for trainer.model.rnnt_conv_transformer_lm."""
from pika_amd.model.prednet import Net  # noqa: F401
