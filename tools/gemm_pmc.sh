#!/bin/bash
# GPU box: SQ counter passes over tools/bin/gemm_tile_ub c (two tile shapes of the GEMM core at a long contraction), per kernel -> stdout
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
CMD="$R/tools/bin/gemm_tile_ub c"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/gpmc$i -o p -- $CMD > /tmp/gpmc$i.log 2>&1 || tail -3 /tmp/gpmc$i.log
done
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/gpmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"gemm_kernel<(\d+), (\d+)", r["Kernel_Name"]); k = m.group(0) if m else None
        if k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in agg.items():
    print(k); [print("   %-28s %.4g per launch" % (c, v / max(cnt[k][c], 1))) for c, v in sorted(d.items())]
PY
