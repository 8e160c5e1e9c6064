"""Drop-in for loader.utt_loader (reference: loader/utt_loader.py)."""
from pika_amd.loader.utt_loader import *  # noqa: F401,F403
from pika_amd.loader.utt_loader import register, dataloader, get_inputdim, splice  # noqa: F401
