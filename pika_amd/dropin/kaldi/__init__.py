"""Minimal stand-in for the PyKaldi names the unchanged PIKA scripts import on the RNN-T path
(SURVEY.md 2.4): CMVN-statistics reading (kaldi.matrix / kaldi.util.io).  Feature extraction,
tables and FSTs are NOT emulated here: the drop-in loaders and decoder do not go through PyKaldi."""
