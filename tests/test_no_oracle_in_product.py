"""The oracle is test infrastructure: nothing under pika_amd/ (or bench.py's timed path) may
import, link or execute it, and the product has no CPU fallback."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_package_never_mentions_the_oracle():
    bad = []
    for path in glob.glob(os.path.join(ROOT, "pika_amd", "**", "*"), recursive=True):
        if not os.path.isfile(path) or not path.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
            continue
        text = open(path, errors="ignore").read()
        if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "liboracle" in text \
                or "oracle/_" in text:
            bad.append(os.path.relpath(path, ROOT))
    assert not bad, "product files reference the oracle: %s" % bad


def test_library_does_not_link_the_oracle():
    import subprocess
    from pika_amd import build
    out = subprocess.run(["ldd", build.build()], capture_output=True, text=True).stdout
    assert "oracle" not in out
