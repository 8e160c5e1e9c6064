"""In-tree build of libpika_amd.so: ``hipcc --offload-arch=gfx950`` on every csrc/*.hip.

No cmake, no JIT cache: the .so sits next to the package so it travels with the repo
snapshot to the GPU box (``*.so`` is git-ignored, not gpurun-ignored).
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libpika_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC]
FLAGS += os.environ.get("PIKA_HIPCC_EXTRA", "").split()      # profiling builds (-DPIKA_ADV_TRACE ...): build with --force
# Per-file code-generation options.  attn.hip: MFMA results in VGPRs instead of AGPRs -- the online softmax reads every
# score and rescales every context accumulator each key tile, so the AGPR form costs ~80 v_accvgpr_read/write per tile
# per wave in kernels that are bound by VALU issue (419 -> 343 instructions in the forward loop, same arithmetic).
EXTRA_FLAGS = {"attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile each .hip to an object (cached by mtime) and link the shared library."""
    objdir = os.path.join(PKG, "_obj")
    os.makedirs(objdir, exist_ok=True)
    headers = glob.glob(os.path.join(INCLUDE, "*.h")) + glob.glob(os.path.join(CSRC, "*.h")) \
        + glob.glob(os.path.join(CSRC, "*.hpp"))
    objs, jobs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, [src, os.path.abspath(__file__)] + headers):
            jobs.append([HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj])
        objs.append(obj)
    if jobs:   # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4, 8)) as ex:
            list(ex.map(run, jobs))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
