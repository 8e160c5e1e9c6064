"""Drop-in for trainer.model.modules.transformer."""
from pika_amd.model.modules import TransformerEncoderLayer  # noqa: F401
