#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the tracked summaries:
profiles/<tag>_kernel_stats.csv, profiles/<tag>_hbm_traffic.md and profiles/traffic.json (read by bench.py for
roofline.traffic).  HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024: MI355X_MICROARCH.md's gfx950
correction (FETCH_SIZE reports half of a wide coalesced read stream); WRITE_SIZE is used as reported."""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
OURS = ["dnn_fwd_kernel", "softmax_ce_kernel", "dnn_bwd_kernel", "dnn_wgrad_kernel", "grad_reduce_kernel",
        "update_kernel", "grad_sumsq_kernel", "click_batch_kernel", "dnn_fb_kernel"]
ALIAS = {"dnn_bwd2_kernel": "dnn_bwd_kernel"}  # the fast row-local backward kernel reports under bench.py's slot name


def short(name):
    for k, v in ALIAS.items():
        if k in name:
            return v
    for k in OURS:
        if k in name:
            return k
    return None


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    if not f:
        sys.exit("missing " + pattern)
    return f[0]


stats = one("stats/**/*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
dst = os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv")
with open(dst, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([r[k] for k in ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]])
avg_us = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows if short(r["Name"])}


def counter(pattern, cname):
    acc, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(one(pattern))):
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == cname:
            acc[k] += float(r["Counter_Value"])
            n[k] += 1
    return {k: acc[k] / n[k] for k in acc}


fetch = counter("fetch/**/*counter_collection.csv", "FETCH_SIZE")
write = counter("write/**/*counter_collection.csv", "WRITE_SIZE")
traffic = {k: (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0 for k in OURS if k in fetch or k in write}
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
with open(os.path.join(ROOT, "profiles", tag + "_hbm_traffic.md"), "w") as fh:
    fh.write("# %s - HBM traffic per launch (rocprofv3 PMC, separate passes), bench.py workload (cfg2: F136 L10 B256 DNN[256,256] IPW)\n\n" % tag)
    fh.write("Collected by `tools/profile_round.sh %s` (`rocprofv3 --pmc FETCH_SIZE --kernel-trace`, then `--pmc WRITE_SIZE\n"
             "--kernel-trace`: one counter per pass, no other trace domains), averaged over all launches of the run.  Counters are\n"
             "in KiB.  Per MI355X_MICROARCH.md (HBM section), on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced\n"
             "read stream, so `traffic` = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes; WRITE_SIZE is used as reported.\n"
             "Durations are from the kernel-trace pass (`%s_kernel_stats.csv`).\n\n" % (tag, tag))
    fh.write("| kernel | avg us | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic bytes/launch |\n|---|---|---|---|---|\n")
    for k in OURS:
        if k in traffic:
            fh.write("| %s | %.2f | %.1f | %.1f | %d |\n" % (k, avg_us.get(k, float("nan")), fetch.get(k, 0), write.get(k, 0), traffic[k]))
print(json.dumps({"avg_us": avg_us, "traffic": traffic}, indent=1))
