#!/bin/bash
# GPU box: SQ counter passes over bench.py's workload, summed per kernel -> stdout
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py ${DNN_PMC_ARGS:---steps 200 --warmup 20} --no-cpu-baseline --no-extras --no-other-configs --spinup-ms 0"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/dpmc$i -o p -- $CMD > /tmp/dpmc$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/dpmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(dnn_\w+|grad_\w+|update_\w+)", r["Kernel_Name"]); k = m.group(1) if m else None
        if k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in agg.items():
    print(k); [print("   %-28s %.4g per launch" % (c, v / max(cnt[k][c], 1))) for c, v in sorted(d.items())]
PY
