"""Generates tests/golden/mbr_script_grads.npz by running the UNCHANGED reference script
/root/reference/trainer/train_transducer_mbr_bmuf_otfaug.py for one batch on the REFERENCE's own modules
(trainer.model.*, decoder.*, trainer.bmuf; CPU, fp32) -- see tests/golden/mbr_hooks.py for the plumbing -- and
recording the N-best its decode produced plus every parameter gradient right before its first optimizer step.
    python tests/golden/make_mbr_script_golden.py
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
out = os.path.join(HERE, "mbr_script_grads.npz")
subprocess.check_call([sys.executable, os.path.join(HERE, "mbr_hooks.py"), "reference", out])
print("wrote", out, os.path.getsize(out) // 1024, "KiB")
