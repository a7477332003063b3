#!/bin/bash
# GPU box: A / B runs of bench.py - ONE script for what rounds 3 - 5 did with a dozen one-liners (ab_cfg / ab_kernels / ab_libs / ab_run /
# env_ab / h3_ab / kernarg_ab / cfg4_fwd_ab / ab_fwd_wide / wgh3_ab: the attempts logs under profiles/ name those).
#   tools/ab.sh [-r REPS] "<bench.py args>" CASE [CASE ...]
# CASE = product                      the built library under the current environment
#      | <variant>                    ultra_pytorch_amd/lib/variants/libultr_<variant>.so (tools/ab_build.sh <variant> "<-D flags>")
#      | VAR=val[,VAR=val...]         the product library under these environment settings (knobs: README)
#      | <variant>:VAR=val[,...]      both
#      | all                          the product library and every built variant
# The cases run round-robin REPS times (default 2: A B A B - boxes drift by 1 - 3 %, compare inside one call); each line: ms/step,
# the kernel averages of the calibration pass, the final loss.
#   tools/ab.sh "--config 5" product ULTR_SR_BLOCK=2          tools/ab.sh -r 3 "--steps 2000" product nopk
cd "$(dirname "$0")/.."
reps=2
if [ "$1" = "-r" ]; then reps=$2; shift 2; fi
args=$1; shift
cases=()
for c in "$@"; do
  if [ "$c" = all ]; then
    cases+=(product)
    for f in ultra_pytorch_amd/lib/variants/libultr_*.so; do [ -f "$f" ] && cases+=("$(basename "$f" .so | sed 's/^libultr_//')"); done
  else cases+=("$c"); fi
done
for rep in $(seq "$reps"); do for c in "${cases[@]}"; do
  var=${c%%:*}; envs=""
  case "$c" in *=*) if [ "$var" = "$c" ]; then var=product; envs=$c; else envs=${c#*:}; fi;; esac
  lib=""
  [ "$var" != product ] && lib="ULTR_HIP_LIB=$PWD/ultra_pytorch_amd/lib/variants/libultr_$var.so"
  printf '%-40s ' "$c"
  env $lib ${envs//,/ } timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-configs $args 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(1e3*d['ms_per_step'],2), d.get('kernel_us'), d.get('final_loss'))"
done; done
