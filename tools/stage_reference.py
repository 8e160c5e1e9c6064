#!/usr/bin/env python
"""Stages the reference's hot-path Python modules for CPU-baseline timing on the GPU box.

BASELINE.md 3 / north_star ask for the reference CPU path "timed on the same box's host cores".  /root/reference does not
exist on the GPU box, but the git-ignored `_ref_scratch/` travels there with the snapshot (like a built .so): this tool
copies the modules of SURVEY 8(a) -- trainer/ (model/, bmuf.py, the MBR script), decoder/, utils/ -- to
`_ref_scratch/reference/`, where oracle/pika_ref.py finds them when /root/reference is absent, and bench.py's cpu_baseline
legs then time the REFERENCE itself, live (tools/time_reference_cpu.py), instead of quoting figures from another host.
Nothing here enters the repository's history, and the product never imports it.

    python tools/stage_reference.py stage | clean | status
"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference"
DST = os.path.join(ROOT, "_ref_scratch", "reference")
PACKAGES = ("trainer", "decoder", "utils")


def stage():
    n = 0
    for pkg in PACKAGES:
        for base, _, files in os.walk(os.path.join(SRC, pkg)):
            for f in files:
                if not f.endswith(".py"):
                    continue
                src = os.path.join(base, f)
                dst = os.path.join(DST, os.path.relpath(src, SRC))
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                shutil.copyfile(src, dst)
                n += 1
    print("staged %d files -> %s" % (n, DST))


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "status"
    if cmd == "stage":
        stage()
    elif cmd == "clean":
        shutil.rmtree(DST, ignore_errors=True)
    print("staged copy present:", os.path.isfile(os.path.join(DST, "trainer", "model", "transducer.py")))
