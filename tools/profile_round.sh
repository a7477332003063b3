#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's workload on the GPU box (run through gpurun from the repo root):
#   pass 1: --kernel-trace --stats            -> per-kernel durations
#   pass 2: --pmc FETCH_SIZE  --kernel-trace  -> HBM read  KiB per dispatch   (one counter per pass, no other domains)
#   pass 3: --pmc WRITE_SIZE  --kernel-trace  -> HBM write KiB per dispatch
#   pass 4: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -> matrix-core busy cycles per dispatch
# Every pass is wrapped in `timeout`; tools/summarize_profiles.py turns gpurun_out/prof_<tag>/ into profiles/.
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-extras --no-other-configs --spinup-ms 0"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- $CMD > "$OUT/stats.log" 2>&1
echo "stats rc=$?"
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o f -- $CMD > "$OUT/fetch.log" 2>&1
echo "fetch rc=$?"
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o w -- $CMD > "$OUT/write.log" 2>&1
echo "write rc=$?"
# MFMA pipe occupancy: SQ counters only (8 slots), again its own pass
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/mfma" -o m -- $CMD > "$OUT/mfma.log" 2>&1
echo "mfma rc=$?"
find "$OUT" -name "*.csv" | head -20
