#!/bin/bash
# Repro matrix for round 3's intermittent wrong result in the split-half dgrad (profiles/r03_cfg2_attempts.md #15).
#   tools/h3_repro.sh build     (build container: cross-compiles the variants into ultra_pytorch_amd/lib/variants/)
#   tools/h3_repro.sh run [reps]  (GPU box: stress every variant; 0 differing launches = deterministic)
# Variants:  A = round 3's first version (epilogue applies the row scale + the cross terms chained on one accumulator set)
#            B = epilogue scale only          C = chained accumulators only        D = the product library
#            AS = A with __syncthreads() for every LDS barrier      AN = A with 32 idle cycles in front of the epilogue
cd "$(dirname "$0")/.."
V=ultra_pytorch_amd/lib/variants
if [ "$1" = build ]; then
  tools/ab_build.sh h3A "-DBWD_H3_EPI=1 -DH3_ACC2=1" h3B "-DBWD_H3_EPI=1" h3C "-DH3_ACC2=1" \
                    h3AS "-DBWD_H3_EPI=1 -DH3_ACC2=1 -DULTR_SYNC_BARRIER=1" h3AN "-DBWD_H3_EPI=1 -DH3_ACC2=1 -DH3_EPI_NOP=1" $EXTRA_VARIANTS
  exit 0
fi
reps=${2:-200}
for v in ${VARIANTS:-h3A h3B h3C h3AS h3AN product}; do
  echo "=== variant $v"
  if [ $v = product ]; then unset ULTR_HIP_LIB; else export ULTR_HIP_LIB=$PWD/$V/libultr_$v.so; fi
  H3_STRESS_ONLY_BWD=1 timeout 300 python tools/h3_stress.py $reps 2>&1 | grep -v amdgpu.ids
done
