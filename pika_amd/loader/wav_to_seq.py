"""wav.scp -> .mrk/.seq converter (reference utils/wav_to_seq.py:11-39, which needs PyKaldi's wave reader).

    python -m pika_amd.loader.wav_to_seq [--num_wav_per_seq 2000] scp:wav.scp out.mrk out.seq

Writes `out.mrk.<i>` / `out.seq.<i>`, a new pair every `num_wav_per_seq` utterances: `.seq` is the
concatenated raw int16 PCM, every `.mrk` line is `uttid byte_offset num_bytes` with offsets restarting at 0 in
each pair -- the container loader/otf_utt_loader.py reads.  wav.scp lines are `uttid /path/to.wav`
(RIFF, 16-bit PCM, mono -- the reference asserts one channel as well); Kaldi command pipes (`... |`) are
not supported here.
"""
import argparse
import wave

import numpy as np


def read_wav_int16(path):
    with wave.open(path, "rb") as w:
        if w.getnchannels() != 1:
            raise ValueError("%s: %d channels, the recipes use mono audio" % (path, w.getnchannels()))
        if w.getsampwidth() != 2:
            raise ValueError("%s: %d-byte samples, expected 16-bit PCM" % (path, w.getsampwidth()))
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")


def iter_wav_scp(rspecifier):
    path = rspecifier.split(":", 1)[1] if rspecifier.startswith(("scp:", "scp,")) else rspecifier
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            p = line.split(None, 1)
            if not p:
                continue
            target = p[1].strip()
            if target.endswith("|"):
                raise NotImplementedError("wav.scp command pipes are not supported: %r" % line.strip())
            yield p[0], read_wav_int16(target)


def convert(rspecifier, out_mrk, out_seq, num_wav_per_seq=2000):
    """Returns the number of (mrk, seq) pairs written."""
    idx = num_written = offset = 0
    mrk = seq = None
    try:
        for uttid, pcm in iter_wav_scp(rspecifier):
            if num_written % num_wav_per_seq == 0:
                if mrk is not None:
                    mrk.close(); seq.close()
                offset = 0
                mrk = open("%s.%d" % (out_mrk, idx), "w", encoding="utf-8")
                seq = open("%s.%d" % (out_seq, idx), "wb")
                idx += 1
            pcm.astype("<i2").tofile(seq)
            mrk.write("{} {} {}\n".format(uttid, offset, 2 * len(pcm)))
            offset += 2 * len(pcm)
            num_written += 1
    finally:
        if mrk is not None:
            mrk.close(); seq.close()
    return idx


def main(argv=None):
    ap = argparse.ArgumentParser(description="wav.scp to seq and mrk file converter")
    ap.add_argument("--num_wav_per_seq", type=int, default=2000)
    ap.add_argument("wav_rspecifier")
    ap.add_argument("out_mrk")
    ap.add_argument("out_seq")
    a, _ = ap.parse_known_args(argv)
    convert(a.wav_rspecifier, a.out_mrk, a.out_seq, a.num_wav_per_seq)


if __name__ == "__main__":
    main()
