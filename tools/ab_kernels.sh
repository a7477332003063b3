#!/bin/bash
# GPU box: per-kernel microseconds of `bench.py <args>` for each library variant (200 steps, all kernel timers):  tools/ab_kernels.sh "<args>" product v1 v2 ...
cd "$(dirname "$0")/.."
args=$1; shift
for v in "$@"; do
  if [ $v = product ]; then unset ULTR_HIP_LIB; else export ULTR_HIP_LIB=$PWD/ultra_pytorch_amd/lib/variants/libultr_$v.so; fi
  echo -n "$v: "; timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 300 $args 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(1e3*d['ms_per_step'],2), d.get('kernel_us'))"
done
