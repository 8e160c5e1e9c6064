"""Drop-in for `from utils.spec_augment import SpecAugment`
(/root/reference/trainer/train_transducer_bmuf_otfaug.py:20)."""
from pika_amd.features import SpecAugment  # noqa: F401
