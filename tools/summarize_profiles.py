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
ALIAS = {"dnn_bwd2_kernel": "dnn_bwd_kernel", "update_tiled_kernel": "update_kernel"}  # the fast row-local backward kernel reports under bench.py's slot name


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
tj_path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
tj.update(traffic)  # (keys this pass did not measure - e.g. SetRank's whole-step sum from tools/summarize_pmc.py - stay)
tj["_source"] = "profiles/%s_hbm_traffic.md (config 2 kernels: tools/profile_round.sh %s)%s" % (
    tag, tag, "; " + tj["_source_setrank"] if "_source_setrank" in tj else "")
json.dump(tj, open(tj_path, "w"), indent=1)
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
    step_kernels = [k for k in ("dnn_fb_kernel", "dnn_wgrad_kernel", "grad_reduce_kernel", "update_kernel") if k in traffic]
    fh.write("\nPer step (%s): **%.1f MB** against 3.06 MB algorithmic (SURVEY 8d).\n" % (" + ".join(step_kernels), sum(traffic[k] for k in step_kernels) / 1e6))
# matrix-core occupancy (optional 4th pass): SQ_VALU_MFMA_BUSY_CYCLES sums, over all SIMDs, the cycles an MFMA occupies its pipe
# (32 per v_mfma_f32_16x16x4_f32); GRBM_GUI_ACTIVE = GPU-busy cycles of the dispatch SUMMED over the 8 XCDs (checked: 8 x the
# kernel duration in shader cycles).  util = busy / (active / 8 * 1024 SIMDs).
mf = glob.glob(os.path.join(src, "mfma/**/*counter_collection.csv"), recursive=True)
if mf:
    busy = counter("mfma/**/*counter_collection.csv", "SQ_VALU_MFMA_BUSY_CYCLES")
    act = counter("mfma/**/*counter_collection.csv", "GRBM_GUI_ACTIVE")
    with open(os.path.join(ROOT, "profiles", tag + "_mfma_util.md"), "w") as fh:
        fh.write("# %s - matrix-core occupancy per launch (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, own pass)\n\n" % tag)
        fb = busy.get("dnn_fb_kernel", 0.0)
        fh.write("`busy` = MFMA pipe cycles summed over the 1024 SIMDs (32 per v_mfma_f32_16x16x4_f32%s),\n"
                 "`active` = GRBM_GUI_ACTIVE of the dispatch, which sums the 8 XCDs (and includes the counter-collection overhead);\n"
                 "utilisation = busy / (active / 8 x 1024).  100 %% = the 157.3 TFLOP/s fp32 MFMA peak.\n"
                 "The figure counts ISSUED MFMAs: the padding rows of the 16-row tiles (10 of 16 live at list_size 10) are in it but\n"
                 "not in bench.py's algorithmic flops.\n\n"
                 % ("; dnn_fb_kernel: %.0f MFMAs per workgroup x 256 workgroups" % (fb / 32.0 / 256.0) if fb else ""))
        fh.write("| kernel | MFMA busy cycles | GPU active cycles (8 XCDs) | matrix-core utilisation |\n|---|---|---|---|\n")
        for k in OURS:
            if k in busy and act.get(k, 0) > 0:
                fh.write("| %s | %.0f | %.0f | %.1f %% |\n" % (k, busy[k], act[k], 100.0 * busy[k] / (act[k] / 8.0 * 1024.0)))
print(json.dumps({"avg_us": avg_us, "traffic": traffic}, indent=1))
