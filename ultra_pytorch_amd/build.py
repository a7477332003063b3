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
# -packed-fp32-ops: hipcc must not select v_pk_{mul,add,fma}_f32.  On gfx950 a packed fp32 instruction whose LOW lane takes the
# HIGH half of a source pair (`op_sel:[0,1]`) intermittently computes that lane with the operand read as zero in lanes 48..63
# while another wave of the same SIMD is executing v_mfma_f32_16x16x32_f16 (tools/pkmul_coexec_test.hip reproduces it in
# isolation: 6 % of executions; profiles/r04_h3_rootcause.md).  hipcc emits that form for `{x, y} * s` wherever s happens to sit
# in the odd register of a pair - round 3's intermittent wrong gradients.  Without packed fp32 the library is as fast
# (profiles/r04_h3_rootcause.md: config 2 48.0 vs 48.3 us, config 3 201 vs 201 us); audit_isa() below enforces it on the binary.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"] + NO_PACKED_FP32
LIBS = []  # no vendor BLAS: every GEMM is the library's own matrix-core code (ultr_gemm.h)
_HOST_NOISE = "'-packed-fp32-ops' is not a recognized feature for this target"  # the x86 half of every compile says so


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
        procs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for cmd, pr in procs:
        err = pr.communicate()[1]
        err = "\n".join(l for l in err.splitlines() if _HOST_NOISE not in l)
        if err.strip():
            print(err, file=sys.stderr)
        if pr.returncode != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + LIBS + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    bad = audit_isa(LIB)
    if bad is None:
        # no disassembler: NOTHING was checked - loud, not fatal (the compile flag NO_PACKED_FP32 is still in force)
        print("WARNING: llvm-objdump not found - the packed-fp32 audit of %s did NOT run" % LIB, file=sys.stderr)
    elif bad:
        os.replace(LIB, LIB + ".rejected")
        raise RuntimeError("libultr_hip.so fails its ISA audit (gfx950 hazard / audit blind, see build.py): " + "; ".join(bad[:5]))
    return LIB


def device_code_objects(path):
    """The gfx950 code objects embedded in a host shared library / object (clang offload bundles), as bytes."""
    blob = open(path, "rb").read()
    magic, out, at = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while True:
        at = blob.find(magic, at)
        if at < 0:
            return out
        n = int.from_bytes(blob[at + 24:at + 32], "little")
        q = at + 32
        for _ in range(n):
            off, size, tlen = (int.from_bytes(blob[q + 8 * k:q + 8 * k + 8], "little") for k in range(3))
            triple = blob[q + 24:q + 24 + tlen].decode("ascii", "replace")
            q += 24 + tlen
            if "gfx950" in triple and size > 0:
                out.append(blob[at + off:at + off + size])
        at += len(magic)


def audit_isa(path=None, objdump=None):
    """Disassemble every gfx950 code object of the library and list the packed fp32 instructions found (none are allowed:
    see NO_PACKED_FP32).  Returns [] when clean, a list of findings otherwise (finding FEWER code objects than translation units
    is a finding: the audit would be blind), None when no disassembler is available (nothing was checked; build_library warns)."""
    import re
    import tempfile
    path = path or LIB
    objdump = objdump or next((c for c in ("/opt/rocm/lib/llvm/bin/llvm-objdump", shutil.which("llvm-objdump")) if c and os.path.exists(c)), None)
    if objdump is None:
        return None
    pat, found = re.compile(r"\bv_pk_(mul|add|fma|min|max)_f32\b|\bv_pk_mov_b32\b"), []
    cos = device_code_objects(path)
    n_units = sum(1 for src in sources() if "__global__" in open(src).read())  # translation units that hold kernels
    if len(cos) < n_units:
        # one gfx950 code object per translation unit is what hipcc embeds today; fewer (compressed bundles - CCOB - or a changed
        # bundle layout) means this audit would look at nothing and pass: say so instead
        return ["audit blind: %d gfx950 code objects found for %d translation units (offload-bundle layout changed?)" % (len(cos), n_units)]
    for k, co in enumerate(cos):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            text = subprocess.run([objdump, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
        sym = "?"
        for line in text.splitlines():
            if line.endswith(">:"):
                sym = line.split("<")[-1][:-2]
            elif pat.search(line):
                found.append("code object %d, %s: %s" % (k, sym, " ".join(line.split()[:6])))
    return found


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
