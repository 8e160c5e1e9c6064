"""Drop-in for trainer.model.modules.multi_headed_attn."""
from pika_amd.model.modules import MultiHeadedAttention  # noqa: F401
