"""Drop-in for trainer.model.modules.position_ffn."""
from pika_amd.model.modules import PositionwiseFeedForward  # noqa: F401
