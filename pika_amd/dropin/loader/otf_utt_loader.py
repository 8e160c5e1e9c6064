"""Drop-in for loader.otf_utt_loader (reference: loader/otf_utt_loader.py)."""
from pika_amd.loader.otf_utt_loader import *  # noqa: F401,F403
from pika_amd.loader.otf_utt_loader import register, dataloader, get_inputdim, splice  # noqa: F401
