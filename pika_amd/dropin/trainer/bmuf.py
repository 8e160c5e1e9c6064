"""Drop-in for `from trainer.bmuf import BmufTrainer`
(/root/reference/trainer/train_transducer_bmuf_otfaug.py:27)."""
from pika_amd.bmuf import BmufTrainer, SUCCESS, STOP  # noqa: F401
