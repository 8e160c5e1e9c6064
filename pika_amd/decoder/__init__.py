"""Batched RNN-T beam search: the search state of ALL utterances
lives in (B, K) device tensors and advances with tensor ops / HIP kernels -- no per-utterance
Python objects, no per-element host reads.  Semantics (including the reference's quirks) follow
decoder/transducer_decoder.py and decoder/beam_transducer.py; see beam_search.py."""
