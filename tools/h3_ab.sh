#!/bin/bash
# GPU box: config 2 with the fused kernel's products on the fp16 matrix cores (split operands, ULTR_FB_H3=1) vs fp32 MFMAs (=0)
for v in 1 0 1 0; do echo "== ULTR_FB_H3=$v"; ULTR_FB_H3=$v timeout 120 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(1e3*d['ms_per_step'],2), d['kernel_us'], d['final_loss'])"; done
