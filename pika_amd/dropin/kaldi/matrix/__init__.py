"""`from kaldi.matrix import _matrix_ext, DoubleMatrix`
(trainer/train_transducer_bmuf_otfaug.py:22, used at :342-344; decoder/decode_transducer.py:10 also imports `Vector`,
which it never uses)."""
import numpy as np

from pika_amd.loader import kaldi_io


class DoubleMatrix(object):
    def __init__(self):
        self.data = np.zeros((0, 0))

    def read_(self, stream, binary):
        """Kaldi's Matrix::Read: the CONTENT decides -- a stream that starts with "\\0B" holds a binary matrix (token DM or
        FM, what `compute-cmvn-stats --binary=true` writes), anything else the text form ` [ ... ]`; the `binary` argument
        (what Input detected) is only a hint, as in Kaldi."""
        self.data = kaldi_io.read_matrix_file(stream.name)
        return self


class _MatrixExt(object):
    @staticmethod
    def double_matrix_to_numpy(m):
        return m.data

    @staticmethod
    def matrix_to_numpy(m):
        return np.asarray(getattr(m, "data", m))


_matrix_ext = _MatrixExt()


class Vector(object):
    """Imported by decode_transducer.py:10 and not used there."""

    def __init__(self, *a, **k):
        raise NotImplementedError("kaldi.matrix.Vector is not on any path of the scripts (SURVEY 2.1)")
