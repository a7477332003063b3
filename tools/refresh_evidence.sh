#!/bin/bash
# GPU box (through gpurun, from the repo root): every tracked bench line + rocprofv3 pass of a round in one call.
#   tools/refresh_evidence.sh r05   ->  gpurun_out/<tag>_*  ; then (here)  tools/refresh_evidence.sh --collect r05  copies into profiles/
if [ "$1" == "--collect" ]; then
  TAG=$2
  for f in gpurun_out/${TAG}_cfg*_bench.json gpurun_out/${TAG}_cfg*_kernel_stats.csv gpurun_out/${TAG}_final_default_bench.json \
           gpurun_out/${TAG}_trace_phases.txt gpurun_out/${TAG}_trace_wide.txt gpurun_out/${TAG}_trace_sr_bwd.txt gpurun_out/${TAG}_dnn_pmc.txt; do [ -s "$f" ] && cp "$f" profiles/; done
  python tools/summarize_profiles.py $TAG
  for C in 3 4pair 5; do [ -d gpurun_out/pmc_${TAG}_cfg$C ] && python tools/summarize_pmc.py $TAG $C > /dev/null; done
  exit 0
fi
TAG=${1:-r05}
mkdir -p gpurun_out
# the driver's own command
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_final_default_bench.json 2> gpurun_out/${TAG}_final_default_bench.err; echo "driver form rc=$?"
for C in 2 3 4pair 4lambda 5; do
  timeout 600 python bench.py --config $C --no-other-configs > gpurun_out/${TAG}_cfg${C}_bench.json 2> gpurun_out/${TAG}_cfg${C}_bench.err; echo "bench cfg $C rc=$?"
done
timeout 600 python bench.py --config 5 --attention-dtype fp16 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_cfg5_fp16_bench.json 2> gpurun_out/${TAG}_cfg5_fp16_bench.err; echo "bench cfg5 fp16 rc=$?"
ULTR_FORCE_DP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-extras --no-other-configs \
  > gpurun_out/${TAG}_cfg2_forced_dp_world1_bench.json 2> gpurun_out/${TAG}_cfg2_forced_dp_world1_bench.err; echo "forced dp rc=$?"
bash tools/profile_round.sh $TAG 2>&1 | grep "rc="
bash tools/profile_configs.sh $TAG 3 4pair 4lambda 5
for C in 3 4pair 5; do bash tools/pmc_config.sh $TAG $C 2>&1 | grep "rc="; done
bash tools/dnn_pmc.sh > gpurun_out/${TAG}_dnn_pmc.txt 2>&1; echo "dnn_pmc rc=$?"
[ -f ultra_pytorch_amd/lib/variants/libultr_trace.so ] || tools/ab_build.sh trace "-DULTR_TRACE" > /dev/null 2>&1
if [ -f ultra_pytorch_amd/lib/variants/libultr_trace.so ]; then
  export ULTR_TRACE_LIB=$PWD/ultra_pytorch_amd/lib/variants/libultr_trace.so
  timeout 300 python tools/trace_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_trace_phases.txt
  for C in 3 4; do
    echo "== config $C: dnn_fwdw_kernel"; timeout 300 python tools/trace_fwdw.py $C 2>&1 | grep -v amdgpu.ids
    echo "== config $C: dnn_bwdw_kernel"; timeout 300 python tools/trace_bwdw.py $C 2>&1 | grep -v amdgpu.ids
  done > gpurun_out/${TAG}_trace_wide.txt
  timeout 300 python tools/trace_sr_bwd.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_trace_sr_bwd.txt
  rm -f ultra_pytorch_amd/lib/variants/libultr_trace.so
fi
echo done
