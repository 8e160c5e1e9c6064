"""Drop-in for decoder.transducer_decoder (reference: decoder/transducer_decoder.py)."""
from pika_amd.decoder.transducer_decoder import TransducerDecoder  # noqa: F401
