"""Drop-in for decoder.beam_transducer: the scripts import GlobalScorer from here
(decoder/decode_transducer.py:15)."""
from pika_amd.decoder.transducer_decoder import GlobalScorer  # noqa: F401
from pika_amd.decoder.beam_search import BeamState as BeamMergeTransducer  # noqa: F401
