"""Drop-in for `from trainer.bmuf import BmufTrainer` (/root/reference/trainer/train_transducer_bmuf_otfaug.py:27)
and `BmufAdamTrainer` (trainer/train_las_bmuf_otfaug.py)."""
from pika_amd.bmuf import BmufAdamTrainer, BmufTrainer, SUCCESS, STOP  # noqa: F401
