"""Drop-in for trainer.model.transducer (reference: trainer/model/transducer.py)."""
from pika_amd.model.transducer import Net  # noqa: F401
