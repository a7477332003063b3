"""Build libultr_hip.so (the gfx950 C-ABI library) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the
resulting ultra_pytorch_amd/lib/libultr_hip.so travels to the GPU box with the snapshot.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libultr_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc"]
LIBS = []  # no vendor BLAS: every GEMM is the library's own matrix-core code (ultr_gemm.h)


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return None


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "ultr_hip.h")]


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(s) <= t for s in deps())


def build_library(force=False, verbose=True):
    """Compile every csrc/*.hip into lib/libultr_hip.so.  Returns the path."""
    if not force and up_to_date():
        return LIB
    cc = _hipcc()
    if cc is None:
        if os.path.exists(LIB):  # GPU box without a toolchain change: use what travelled
            return LIB
        raise RuntimeError("hipcc not found and %s does not exist" % LIB)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [cc] + FLAGS + sources() + LIBS + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
