"""`import kaldi.fstext as fst` (decoder/decode_transducer.py:11, used at :83 `fst.StdVectorFst.read(args.fst_lm)`):
the n-gram LM is read from its OpenFST binary file into the OpenFST-free CSR table the drop-in SortedMatcher
(decoder/sorted_matcher.py) searches."""
from pika_amd.decoder.ngram_fst import NgramFst


class StdVectorFst(object):
    @staticmethod
    def read(path):
        return NgramFst.read_binary(path)
