REPO=$(pwd)
for m in "0 -1" "1 10" "1 15"; do set -- $m; echo -n "ATTN_H3=$1 MASK=$2: "; ULTR_SR_ATTN_H3=$1 ULTR_SR_ATTN_H3_MASK=$2 python bench.py --config 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(1e3*d['ms_per_step'],1))"; done
cd /tmp && export TMPDIR=/tmp
ULTR_SR_ATTN_H3=1 ULTR_SR_ATTN_H3_MASK=15 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ah3 -o s -- python $REPO/bench.py --config 5 --no-cpu-baseline --no-extras --steps 20 > /tmp/ah3.log 2>&1
python - <<PY
import csv,glob
fs=glob.glob('/tmp/ah3/**/*kernel_stats.csv',recursive=True)
for r in csv.DictReader(open(fs[0])):
    if 'attn' in r['Name']: print('   %-60s calls %4s avg %7.1f us'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
