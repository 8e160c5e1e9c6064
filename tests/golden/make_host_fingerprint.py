"""Records how THIS host's fp32 CPU library sums (tests/golden/host_fingerprint.json): the bits of a seeded 96 x 1024 x 96
product and of a log-softmax over 5000 values.  The full-width CPU decode test demands bit-identical n-best lists when the
test host sums like the host that recorded the reference golden, and applies its noise criterion only elsewhere
(ADVICE r4).  Run in the container that (re)generates tests/golden/decode_full.npz:
    python tests/golden/make_host_fingerprint.py
"""
import hashlib
import json
import os

import torch


def fingerprint():
    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(20240926)
    a = torch.randn(96, 1024, generator=g)
    b = torch.randn(1024, 96, generator=g)
    v = torch.randn(7, 5000, generator=g) * 9
    h = hashlib.sha256()
    h.update((a @ b).numpy().tobytes())
    h.update(torch.log_softmax(v, dim=-1).numpy().tobytes())
    h.update(torch.tanh(a[:4]).numpy().tobytes())
    return h.hexdigest()


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_fingerprint.json")
    json.dump({"fp32_cpu_sha256": fingerprint(), "torch": torch.__version__}, open(path, "w"))
    print(open(path).read())
