#!/bin/bash
# GPU box: bench.py with each library variant in turn, twice (A B A B):  tools/ab_libs.sh "<bench args>" product nopk ...
cd "$(dirname "$0")/.."
args=$1; shift
for rep in 1 2; do for v in "$@"; do
  if [ $v = product ]; then unset ULTR_HIP_LIB; else export ULTR_HIP_LIB=$PWD/ultra_pytorch_amd/lib/variants/libultr_$v.so; fi
  echo -n "$v: "; timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(1e3*d['ms_per_step'],2), d.get('kernel_us'), d['final_loss'])"
done; done
