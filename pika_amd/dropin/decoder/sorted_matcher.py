"""Drop-in for decoder.sorted_matcher (reference: decoder/sorted_matcher.py)."""
from pika_amd.decoder.ngram_fst import SortedMatcher, NgramFst  # noqa: F401
