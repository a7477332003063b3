#!/bin/bash
# GPU box (through gpurun, from the repo root): rocprofv3 counter passes for `bench.py --config C` - every kernel of the step.
#   tools/pmc_config.sh r02 4pair [extra bench flags]  ->  gpurun_out/pmc_<tag>_cfg<C>/{stats,fetch,write,mfma}; then (here)
#   python tools/summarize_pmc.py r02 4pair  ->  profiles/<tag>_cfg<C>_pmc.md
# One counter set per pass, kernel-trace only (no other trace domains), every pass under `timeout`.
TAG=$1; C=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_${TAG}_cfg$C
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CC=$C; EXTRA="$@"
if [ "$C" == "5fp16" ]; then CC=5; EXTRA="--attention-dtype fp16 $EXTRA"; fi
CMD="python $REPO/bench.py --config $CC $EXTRA --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- $CMD > "$OUT/stats.log" 2>&1; echo "stats rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o f -- $CMD > "$OUT/fetch.log" 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o w -- $CMD > "$OUT/write.log" 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/mfma" -o m -- $CMD > "$OUT/mfma.log" 2>&1; echo "mfma rc=$?"
