#!/bin/bash
# GPU box: SQ counter passes over a few SetRank steps (tools/setrank_prof.py), summed per kernel -> stdout
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
export SR_B=${SR_B:-256}
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/srpmc$i -o p -- python $R/tools/setrank_prof.py > /tmp/srpmc$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob("/tmp/srpmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(sr_\w+|Cijk_\w{0,24})", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    if "attn" in k or "wgrad" in k:
        print(k); [print("   %-28s %.4g" % (c, v)) for c, v in sorted(d.items())]
PY
