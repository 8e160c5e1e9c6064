"""wav.scp -> byte-count list (reference utils/wav_to_bytes.py:9-24, which needs PyKaldi's wave reader).

    python -m pika_amd.loader.wav_to_bytes scp:wav.scp out.bytes

Every line of `out.bytes` is `uttid num_bytes` with num_bytes = 2 x samples of the mono 16-bit signal: the length
file utils/shuffle_by_length.py and utils/split_by_length.py group utterances by."""
import argparse

from .wav_to_seq import iter_wav_scp


def convert(rspecifier, byte_file):
    n = 0
    with open(byte_file, "w") as bf:
        for uttid, pcm in iter_wav_scp(rspecifier):
            bf.write("{} {}\n".format(uttid, 2 * len(pcm)))
            n += 1
    return n


def main(argv=None):
    ap = argparse.ArgumentParser(description="length list of a wav.scp: one `uttid bytes` line per utterance")
    ap.add_argument("wav_rspecifier")
    ap.add_argument("byte_file")
    args, _ = ap.parse_known_args(argv)
    convert(args.wav_rspecifier, args.byte_file)


if __name__ == "__main__":
    main()
