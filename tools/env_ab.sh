#!/bin/bash
# GPU box: bench one config under several environment settings:  tools/env_ab.sh 3 "ULTR_FWD_H3=0" "ULTR_FWD_H3=1" ...
c=$1; shift
for e in "$@"; do
  for rep in 1 2; do
    echo -n "config $c  $e : "
    env $e timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(1e3*d['ms_per_step'],2), d['kernel_us'])"
  done
done
