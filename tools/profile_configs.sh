#!/bin/bash
# rocprofv3 kernel statistics of `bench.py --config C` for the other BASELINE configs (run through gpurun from the repo root):
#   tools/profile_configs.sh r02 3 4pair 4lambda 5 5fp16   ->  gpurun_out/prof_<tag>_cfg<C>/ ; copy *kernel_stats.csv into profiles/
TAG=$1; shift
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for C in "$@"; do
  OUT=$REPO/gpurun_out/prof_${TAG}_cfg$C
  rm -rf "$OUT"; mkdir -p "$OUT"
  EXTRA=""; CC=$C
  if [ "$C" == "5fp16" ]; then CC=5; EXTRA="--attention-dtype fp16"; fi  # config 5 with the opt-in fp16-operand attention
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- \
    python $REPO/bench.py --config $CC $EXTRA --no-cpu-baseline --no-extras > "$OUT/stats.log" 2>&1
  echo "cfg $C rc=$?"
  f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$REPO/gpurun_out/${TAG}_cfg${C}_kernel_stats.csv"
done
