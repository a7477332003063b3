R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|SQC_INST|INST_LEVEL" | head -40
CMD="python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-other-configs --spinup-ms 0"
timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d /tmp/ic1 -o p -- $CMD > /tmp/ic1.log 2>&1
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/ic1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(dnn_\w+|grad_\w+|update_\w+)", r["Kernel_Name"]); k = m.group(1) if m else None
        if k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in agg.items():
    print(k); [print("   %-28s %.4g per launch" % (c, v / max(cnt[k][c], 1))) for c, v in sorted(d.items())]
PY
tail -3 /tmp/ic1.log
