# GPU box: per-kernel durations of the per-layer forward (ULTR_BIG_FWD=2) at configs 4 and 3
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in 4pair 3; do
ULTR_BIG_FWD=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_big_$c -o s -- python $REPO/bench.py --config $c --no-cpu-baseline --no-extras --steps 50 > /tmp/prof_big_$c.log 2>&1
python - <<PY
import csv,glob
fs=glob.glob('/tmp/prof_big_$c/**/*kernel_stats.csv',recursive=True)
if not fs:
    print(open('/tmp/prof_big_$c.log').read()[-2000:])
else:
    rows=list(csv.DictReader(open(fs[0])))
    print('cfg $c')
    for r in rows[:14]: print('  %-90s calls %5s avg %8.1f us'%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
