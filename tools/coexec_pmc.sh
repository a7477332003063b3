#!/bin/bash
# GPU box: do VALU and MFMA instructions of different waves overlap on a SIMD?  SQ_VALU_MFMA_COEXEC_CYCLES next to the busy counters,
# per kernel of `bench.py --config C`:  tools/coexec_pmc.sh 3
R=$GRAFT_REPO_ROOT; C=${1:-3}; cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --config $C --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-other-configs --spinup-ms 0"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MFMA SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/cpmc$i -o p -- $CMD > /tmp/cpmc$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/cpmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(dnn_\w+|grad_\w+|update_\w+|sr_\w+|gemm_h3_kernel)", r["Kernel_Name"]); k = m.group(1) if m else None
        if k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in agg.items():
    print(k); [print("   %-28s %.4g per launch" % (c, v / max(cnt[k][c], 1))) for c, v in sorted(d.items())]
PY
