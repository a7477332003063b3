#!/bin/bash
# GPU box: configs 3 / 4 with and without the wide-tile forward (ULTR_FWD_WIDE), step time + the kernels of the step
for c in ${1:-3 4pair}; do
  for w in 1 0; do
    echo "== config $c  ULTR_FWD_WIDE=$w"
    ULTR_FWD_WIDE=$w timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras --no-other-configs 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(1e3*d['ms_per_step'],2), d['kernel_us'])"
  done
done
