"""Registers, spills and LDS of every kernel in the built library (from the code objects' metadata notes).
usage: python tools/kernel_resources.py [substring ...]"""
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, ".")
from ultra_pytorch_amd import build  # noqa: E402

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FIELDS = ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")


def kernels(path=None):
    out = []
    for co in build.device_code_objects(path or build.LIB):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            text = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        cur = {}
        for line in text.splitlines():
            m = re.match(r"\s+(- )?\.(\w+):\s+(.*)", line)
            if not m:
                continue
            if m.group(1) and m.group(2) in ("agpr_count", "args") and cur.get("name"):
                out.append(cur)
                cur = {}
            if m.group(2) in FIELDS:
                cur[m.group(2)] = m.group(3).strip("'")
        if cur.get("name"):
            out.append(cur)
    return out


if __name__ == "__main__":
    pats = sys.argv[1:]
    for k in kernels():
        if pats and not any(p in k["name"] for p in pats):
            continue
        print("%-100s vgpr %3s agpr %3s sgpr %3s spill %3s scratch %4s lds %6s" % (
            k["name"][:100], k.get("vgpr_count"), k.get("agpr_count", "0"), k.get("sgpr_count"), k.get("vgpr_spill_count", "0"),
            k.get("private_segment_fixed_size", "0"), k.get("group_segment_fixed_size", "0")))
