"""Drop-in for trainer.model.las (inference scoring path; reference: trainer/model/las.py)."""
from pika_amd.model.las import (Net, LASRNNEncoder, LASEmbeddings, InputFeedRNNDecoder,  # noqa: F401
                                score_nbest_batch_many)
