"""`importlib.import_module("model." + proto)` target of the training scripts
(/root/reference/trainer/train_transducer_bmuf_otfaug.py:293): same class as trainer.model.transducer."""
from pika_amd.model.transducer import Net  # noqa: F401
