"""Build recipe of libfrcnn_hip.so (gfx950 only): plain `hipcc -c` per translation unit, then one link.

The library is built IN-TREE (``faster_rcnn_pytorch_multimodal_amd/lib/libfrcnn_hip.so``) so it travels
with a snapshot of the repository to a GPU box; nothing is JIT-compiled at import time.
"""
import hashlib
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OUT_DIR = os.path.join(PKG_DIR, "lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
LIB_PATH = os.path.join(OUT_DIR, "libfrcnn_hip.so")
ARCH = "gfx950"

# translation unit -> extra flags.  The box-arithmetic units must not contract mul+add into fma: the
# reference evaluates those expressions as separate elementwise torch ops (one rounding each).
SOURCES = {
    "common.hip": [],
    "conv_igemm.hip": [],
    "conv_wgrad.hip": [],
    "boxes.hip": ["-ffp-contract=off"],
    "roi_align.hip": ["-ffp-contract=off"],
    "head.hip": ["-ffp-contract=off"],
    "train_ops.hip": ["-ffp-contract=off"],
    "targets.hip": ["-ffp-contract=off"],
    "prep.hip": ["-ffp-contract=off"],
    "batchnorm.hip": ["-ffp-contract=off"],
    "voxelize.hip": ["-ffp-contract=off"],
}
COMMON_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC); libfrcnn_hip.so cannot be built")


def _stamp(src, flags):
    h = hashlib.sha1()
    h.update(" ".join(COMMON_FLAGS + flags).encode())
    for dep in [src] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + [
        os.path.join(PKG_DIR, "..", "include", "frcnn_hip.h")
    ]:
        with open(dep, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link libfrcnn_hip.so. Returns the library path."""
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs, rebuilt = [], False
    for name, flags in SOURCES.items():
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ_DIR, name.replace(".hip", ".o"))
        stamp_file = obj + ".stamp"
        stamp = _stamp(src, flags)
        fresh = (not force and os.path.exists(obj) and os.path.exists(stamp_file)
                 and open(stamp_file).read() == stamp)
        if not fresh:
            cmd = [hipcc] + COMMON_FLAGS + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
            with open(stamp_file, "w") as f:
                f.write(stamp)
            rebuilt = True
        objs.append(obj)
    if rebuilt or not os.path.exists(LIB_PATH):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
