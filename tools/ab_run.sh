#!/bin/bash
# GPU box: bench every variant built by tools/ab_build.sh (plus the product library), print ms/step and kernel times
for f in ultra_pytorch_amd/lib/libultr_hip.so ultra_pytorch_amd/lib/variants/*.so; do
  echo "== $f"
  ULTR_HIP_LIB=$PWD/$f timeout 120 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(1e3*d['ms_per_step'],2), d['kernel_us'])"
done
