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
OBJ = os.path.join(HERE, "lib", "obj")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]
LIBS = []  # no vendor BLAS: every GEMM is the library's own matrix-core code (ultr_gemm.h)


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return None


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "ultr_hip.h")]


def deps():
    return sources() + headers()


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
    os.makedirs(OBJ, exist_ok=True)
    # one object per source, compiled in parallel, rebuilt only when the source or any header is newer
    hdr_t = max(os.path.getmtime(h) for h in headers())
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([cc] + CFLAGS + ["-c", src, "-o", obj])
    procs = []
    for cmd in jobs:
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + LIBS + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
