#!/bin/bash
# GPU box: bench the product library and every variant (tools/ab_build.sh) at the given configs:  tools/ab_cfg.sh "3 4pair 5"
for c in $1; do
  for f in ultra_pytorch_amd/lib/libultr_hip.so ultra_pytorch_amd/lib/variants/*.so; do
    [ -f "$f" ] || continue
    echo "== config $c  $f"
    ULTR_HIP_LIB=$PWD/$f timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(1e3*d['ms_per_step'],2), d['kernel_us'])"
  done
done
