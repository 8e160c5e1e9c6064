"""`from kaldi.util import io` -> `io.Input(path, binary=False)` context manager
(trainer/train_transducer_bmuf_otfaug.py:23,342)."""


class Input(object):
    def __init__(self, path, binary=False):
        self.path, self.binary = path, binary
        self._f = None

    def __enter__(self):
        # Kaldi's Input opens the file and LOOKS: `binary` becomes what the content says ("\0B" header), whatever the caller
        # passed (train_transducer_bmuf_otfaug.py:342 passes binary=False and then hands ki.binary to Matrix.read_)
        with open(self.path, "rb") as f:
            self.binary = f.read(2) == b"\0B"
        self._f = open(self.path, "rb" if self.binary else "r")
        return self

    def __exit__(self, *exc):
        self._f.close()

    def stream(self):
        return self._f
