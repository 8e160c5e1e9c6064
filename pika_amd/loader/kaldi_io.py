"""Readers for the on-disk formats that feed the hot path, without PyKaldi:

* `.mrk` / `.seq` audio containers (utils/wav_to_seq.py:28-39): `.mrk` line = `uttid byte_offset
  num_bytes`, `.seq` = concatenated raw int16 PCM;
* Kaldi int-vector archives (labels): text `uttid i1 i2 ...` or binary (`uttid \\0B` + int32 count
  with 1-byte size markers), read specifiers `ark:path`, `ark,t:path` or a plain path;
* Kaldi float-matrix archives / scp (offline features for loader/utt_loader.py), plain and compressed (CM / CM2 / CM3);
* Kaldi text matrices (the CMVN statistics file, train_transducer_bmuf_otfaug.py:341-346).
"""
import struct

import numpy as np


def _rspec_path(rspec):
    if ":" in rspec and rspec.split(":", 1)[0].replace(",", "").isalpha():
        kind, path = rspec.split(":", 1)
        return kind, path
    return "ark", rspec


def read_mrk(path):
    """[(uttid, byte_offset, num_bytes)]"""
    out = []
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            p = line.split()
            if p:
                out.append((p[0], int(p[1]), int(p[2])))
    return out


def read_pcm(seq_file, offset, num_bytes):
    num_bytes -= num_bytes % 2                      # otf_utt_loader.py:215
    seq_file.seek(offset)
    return np.frombuffer(seq_file.read(num_bytes), dtype="int16")


def _read_token(f):
    tok = b""
    while True:
        c = f.read(1)
        if not c:
            return None
        if c in b" \n\t":
            if tok:
                return tok.decode("utf-8")
            continue
        tok += c


def _read_int_vector_body(f):
    """One int vector at the stream position, behind its key: binary `\\0B <4> <int32 n>` + n x (<4> <int32>), or the rest
    of a text line (an optional `[ ... ]` is tolerated: copy-int-vector writes none, some tools do)."""
    pos = f.tell()
    head = f.read(2)
    if head == b"\0B":
        assert f.read(1) == b"\x04"
        n = struct.unpack("<i", f.read(4))[0]
        raw = np.frombuffer(f.read(5 * n), dtype=np.uint8).reshape(n, 5)
        return raw[:, 1:].copy().view("<i4").reshape(n).astype(np.int32)
    f.seek(pos)
    parts = [v for v in f.readline().decode("utf-8").split() if v not in ("[", "]")]
    return np.array([int(v) for v in parts], np.int32)


def read_int_vectors(rspec):
    """Yield (uttid, np.int32 array) from a Kaldi int-vector table: an archive `ark:file` / `ark,t:file` / a plain path
    (text or binary, decided per entry by its content), or a script file `scp:file` whose lines are
    `uttid path[:byte_offset]` -- the offset points BEHIND the key of an archive entry, as `copy-int-vector ark:.. ark,scp:..`
    writes it; without an offset the file holds one vector."""
    kind, path = _rspec_path(rspec)
    if kind.startswith("scp"):
        with open(path, "r", encoding="utf-8") as s:
            for line in s:
                p = line.split(None, 1)
                if not p:
                    continue
                loc = p[1].strip()
                fn, off = loc.rsplit(":", 1) if (":" in loc and loc.rsplit(":", 1)[1].isdigit()) else (loc, "0")
                with open(fn, "rb") as f:
                    f.seek(int(off))
                    yield p[0], _read_int_vector_body(f)
        return
    with open(path, "rb") as f:
        while True:
            pos = f.tell()
            key = _read_token(f)
            if key is None:
                return
            f.seek(-1, 1)
            if f.read(1) == b"\n":                  # `uttid\n`: a text entry with no labels (the token ate its line end)
                yield key, np.zeros(0, np.int32)
                continue
            yield key, _read_int_vector_body(f)


def _read_compressed_matrix(f, tok):
    """Kaldi's CompressedMatrix on disk (what `copy-feats --compress=true` writes; the archives loader/utt_loader.py:163-164
    reads through PyKaldi).  After the token -- "CM" (format 1), "CM2", "CM3" -- come the remaining 16 bytes of the global
    header as one raw struct {float min_value, float range, int32 rows, int32 cols} (no size markers), then
      CM : per column four uint16 percentiles (0, 25, 75, 100 %; value = min + range * q / 65535), then one byte per
           element COLUMN by column: three linear pieces  0..64 -> [p0, p25], 64..192 -> [p25, p75], 192..255 -> [p75, p100];
      CM2: uint16 per element, row-major: min + range * q / 65535;
      CM3: one byte per element, row-major: min + range * q / 255.
    Arithmetic in float32 in Kaldi's order of operations, so the floats are the ones its reader produces."""
    lo, rng, rows, cols = struct.unpack("<ffii", f.read(16))
    lo, rng = np.float32(lo), np.float32(rng)
    if rows < 0 or cols < 0:
        raise ValueError("corrupt compressed Kaldi matrix header (%d x %d)" % (rows, cols))
    if tok == "CM":
        heads = np.frombuffer(f.read(8 * cols), dtype="<u2").reshape(cols, 4).astype(np.float32)
        q = np.frombuffer(f.read(rows * cols), dtype=np.uint8)
        if q.size != rows * cols:
            raise ValueError("truncated compressed Kaldi matrix")
        q = q.reshape(cols, rows).astype(np.float32)
        pc = lo + rng * np.float32(1.0 / 65535.0) * heads                         # (cols, 4)
        p0, p25, p75, p100 = (pc[:, i:i + 1] for i in range(4))
        out = np.where(q <= 64, p0 + (p25 - p0) * q * np.float32(1 / 64.0),
                       np.where(q <= 192, p25 + (p75 - p25) * (q - 64) * np.float32(1 / 128.0),
                                p75 + (p100 - p75) * (q - 192) * np.float32(1 / 63.0)))
        return np.ascontiguousarray(out.T.astype(np.float32))
    if tok == "CM2":
        q = np.frombuffer(f.read(2 * rows * cols), dtype="<u2")
        inc = rng * np.float32(1.0 / 65535.0)
    else:
        q = np.frombuffer(f.read(rows * cols), dtype=np.uint8)
        inc = rng * np.float32(1.0 / 255.0)
    if q.size != rows * cols:
        raise ValueError("truncated compressed Kaldi matrix")
    return (lo + q.reshape(rows, cols).astype(np.float32) * inc).astype(np.float32)


def _read_binary_matrix(f):
    head = f.read(2)
    assert head == b"\0B", "expected a binary Kaldi matrix"
    tok = _read_token(f)
    if tok in ("CM", "CM2", "CM3"):
        return _read_compressed_matrix(f, tok)
    if tok not in ("FM", "DM"):
        raise NotImplementedError("Kaldi matrix type %r" % tok)
    assert f.read(1) == b"\x04"
    rows = struct.unpack("<i", f.read(4))[0]
    assert f.read(1) == b"\x04"
    cols = struct.unpack("<i", f.read(4))[0]
    dt = "<f4" if tok == "FM" else "<f8"
    data = np.frombuffer(f.read(rows * cols * int(dt[-1])), dtype=dt).reshape(rows, cols)
    return data.astype(np.float32)


def read_matrices(rspec):
    """Yield (uttid, float32 matrix) from `ark:file` (binary) or `scp:file` (path:offset lines)."""
    kind, path = _rspec_path(rspec)
    if kind.startswith("scp"):
        with open(path, "r", encoding="utf-8") as s:
            for line in s:
                key, loc = line.split(None, 1)
                fn, off = loc.strip().rsplit(":", 1) if ":" in loc else (loc.strip(), "0")
                with open(fn, "rb") as f:
                    f.seek(int(off))
                    yield key, _read_binary_matrix(f)
        return
    with open(path, "rb") as f:
        while True:
            key = _read_token(f)
            if key is None:
                return
            yield key, _read_binary_matrix(f)


def write_matrix_ark(path, items):
    """Binary float-matrix archive writer (tests and data prep)."""
    with open(path, "wb") as f:
        for key, m in items:
            m = np.ascontiguousarray(m, dtype="<f4")
            f.write(key.encode("utf-8") + b" \0BFM \x04" + struct.pack("<i", m.shape[0]) + b"\x04" +
                    struct.pack("<i", m.shape[1]) + m.tobytes())


def read_matrix_file(path):
    """A single Kaldi matrix file, text or binary by its content (Matrix::Read): float64 array.  Binary: `\\0B` + token `DM`
    (double) or `FM` (float) + <4> rows <4> cols + row-major data; compressed forms as in _read_compressed_matrix."""
    with open(path, "rb") as f:
        if f.read(2) != b"\0B":
            return read_text_matrix(path)
        tok = _read_token(f)
        if tok in ("CM", "CM2", "CM3"):
            return _read_compressed_matrix(f, tok).astype(np.float64)
        if tok not in ("FM", "DM"):
            raise NotImplementedError("Kaldi matrix type %r in %s" % (tok, path))
        assert f.read(1) == b"\x04"
        rows = struct.unpack("<i", f.read(4))[0]
        assert f.read(1) == b"\x04"
        cols = struct.unpack("<i", f.read(4))[0]
        dt = "<f4" if tok == "FM" else "<f8"
        data = np.frombuffer(f.read(rows * cols * int(dt[-1])), dtype=dt)
        if data.size != rows * cols:
            raise ValueError("truncated Kaldi matrix in %s" % path)
        return data.reshape(rows, cols).astype(np.float64)


def read_text_matrix(path):
    """Kaldi text matrix ` [ a b c\\n d e f ]` -> float64 array (CMVN statistics: 2 x (D+1))."""
    with open(path, "r") as f:
        txt = f.read()
    body = txt[txt.index("[") + 1: txt.rindex("]")]
    rows = [[float(v) for v in line.split()] for line in body.strip().split("\n") if line.strip()]
    return np.array(rows, np.float64)


def cmvn_offset_scale(stats, repeat=1, floor=1.0e-20):
    """mean/var from accumulated stats -> (offset, scale), each repeated over the splice window
    (train_transducer_bmuf_otfaug.py:345-355)."""
    mean = stats[0][:-1] / stats[0][-1]
    var = stats[1][:-1] / stats[0][-1] - mean * mean
    if np.min(np.abs(var)) < floor:
        raise ValueError("problematic cmvn_stats, variance too small")
    return np.tile(-mean, repeat), np.tile(1.0 / np.sqrt(var), repeat)


def write_int_vectors(path, items, binary=False):
    """Kaldi int-vector archive (label alignments): text `uttid i1 i2 ...` or binary
    (`uttid \\0B` + 1-byte size marker 4 + int32 count + per element marker 4 + int32)."""
    if not binary:
        with open(path, "w", encoding="utf-8") as f:
            for key, vec in items:
                # Kaldi's text holder: key, a space, the elements each FOLLOWED by a space, the line end (`utt1 \n` when empty)
                f.write(key + " " + "".join("%d " % int(v) for v in vec) + "\n")
        return
    with open(path, "wb") as f:
        for key, vec in items:
            f.write(key.encode("utf-8") + b" \0B")
            f.write(struct.pack("<bi", 4, len(vec)))
            for v in vec:
                f.write(struct.pack("<bi", 4, int(v)))


def write_text_matrix(path, mat):
    """Kaldi text matrix ` [\n  r0 ...\n  r1 ... ]` (the CMVN statistics file format)."""
    mat = np.asarray(mat, dtype=np.float64)
    with open(path, "w", encoding="utf-8") as f:
        f.write(" [\n")
        for i, row in enumerate(mat):
            f.write("  " + " ".join(repr(float(v)) for v in row) + (" ]\n" if i == len(mat) - 1 else "\n"))


def accumulate_cmvn_stats(feature_iter):
    """2 x (D+1) Kaldi CMVN statistics (row 0: sums + frame count, row 1: sums of squares + 0) from an iterable
    of (T, D) feature matrices -- what `compute-cmvn-stats` writes and cmvn_offset_scale() consumes."""
    stats = None
    for feats in feature_iter:
        x = np.asarray(feats, dtype=np.float64)
        if stats is None:
            stats = np.zeros((2, x.shape[1] + 1))
        stats[0, :-1] += x.sum(0)
        stats[1, :-1] += (x * x).sum(0)
        stats[0, -1] += x.shape[0]
    return stats

