#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>_cfg<C>/ (tools/pmc_config.sh) -> profiles/<tag>_cfg<C>_pmc.md: every kernel of the step with its
average duration (kernel-trace pass), HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (MI355X_MICROARCH.md's
gfx950 correction: FETCH_SIZE reports half of a wide coalesced read stream; WRITE_SIZE as reported; separate passes) and the
matrix-core occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
import csv, glob, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cfg = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "pmc_%s_cfg%s" % (tag, cfg))


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    return re.sub(r"\s+", " ", n)[:96]


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    if not f:
        sys.exit("missing " + pattern)
    return f[0]


dur, calls = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(one("stats/**/*kernel_trace.csv"))):
    k = short(r["Kernel_Name"])
    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    calls[k] += 1


def counter(pattern, cname):
    acc, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(one(pattern))):
        if r["Counter_Name"] == cname:
            k = short(r["Kernel_Name"])
            acc[k] += float(r["Counter_Value"])
            n[k] += 1
    return {k: acc[k] / n[k] for k in acc}


fetch = counter("fetch/**/*counter_collection.csv", "FETCH_SIZE")
write = counter("write/**/*counter_collection.csv", "WRITE_SIZE")
busy = counter("mfma/**/*counter_collection.csv", "SQ_VALU_MFMA_BUSY_CYCLES")
act = counter("mfma/**/*counter_collection.csv", "GRBM_GUI_ACTIVE")
steps = max(calls.get(k, 0) for k in calls if "update" in k)
total = sum(dur.values())
out = os.path.join(ROOT, "profiles", "%s_cfg%s_pmc.md" % (tag, cfg))
with open(out, "w") as fh:
    fh.write("# %s - `bench.py --config %s`: every kernel of the training step (rocprofv3, separate passes)\n\n" % (tag, cfg))
    fh.write("`tools/pmc_config.sh %s %s`: pass 1 `--kernel-trace --stats` (durations), pass 2 `--pmc FETCH_SIZE`, pass 3 `--pmc WRITE_SIZE`,\n"
             "pass 4 `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` (kernel-trace only beside the counters).  HBM bytes per\n"
             "launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950: FETCH_SIZE reports half of a wide coalesced read stream);\n"
             "matrix-core occupancy = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), 100 %% = the 157.3 TFLOP/s fp32 peak (fp16\n"
             "MFMAs occupy the same pipe).  %d steps in the run; launches per step from the trace.\n\n" % (tag, cfg, steps))
    fh.write("| kernel | launches / step | avg us | share of step | HBM MB / launch | HBM TB/s | matrix-core occupancy |\n|---|---|---|---|---|---|---|\n")
    for k in sorted(dur, key=lambda k: -dur[k]):
        if calls[k] < steps // 2:
            continue
        us = dur[k] / calls[k]
        b = (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0
        occ = 100.0 * busy[k] / (act[k] / 8.0 * 1024.0) if k in busy and act.get(k, 0) > 0 else float("nan")
        fh.write("| `%s` | %.1f | %.1f | %.1f %% | %.1f | %.2f | %s |\n" % (k, calls[k] / steps, us, 100.0 * dur[k] / total, b / 1e6, b / us / 1e6,
                                                                       ("%.0f %%" % occ) if occ == occ else "-"))
if cfg.startswith("5"):
    # the whole step's HBM bytes (sum over its launches) -> profiles/traffic.json, read by bench.py for config 5's roofline.traffic
    import json
    tot_b = sum((2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0 * calls[k] / steps for k in dur if calls[k] >= steps // 2)
    tj_path = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
    if cfg == "5":
        tj["setrank_whole_step"] = tot_b
        tj["_source_setrank"] = "profiles/%s_cfg5_pmc.md (SetRank: sum over the launches of one step)" % tag
        json.dump(tj, open(tj_path, "w"), indent=1)
    with open(out, "a") as fh:
        fh.write("\nWhole step: **%.2f GB** of HBM traffic over %d launches, %.0f us of kernels.\n"
                 % (tot_b / 1e9, round(sum(calls[k] / steps for k in dur if calls[k] >= steps // 2)), sum(dur[k] / steps for k in dur if calls[k] >= steps // 2)))
print(open(out).read())
