"""`from kaldi.util import io` -> `io.Input(path, binary=False)` context manager
(trainer/train_transducer_bmuf_otfaug.py:23,342)."""


class Input(object):
    def __init__(self, path, binary=False):
        self.path, self.binary = path, binary
        self._f = None

    def __enter__(self):
        self._f = open(self.path, "rb" if self.binary else "r")
        return self

    def __exit__(self, *exc):
        self._f.close()

    def stream(self):
        return self._f
