"""Steps immediately after decode: n-best reranking and scoring, Kaldi-free."""
