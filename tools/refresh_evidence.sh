#!/bin/bash
# GPU box (through gpurun, from the repo root): every tracked bench line + rocprofv3 pass of a round in one call.
#   tools/refresh_evidence.sh r02   ->  gpurun_out/<tag>_*  ; then (here)  tools/refresh_evidence.sh --collect r02  copies into profiles/
if [ "$1" == "--collect" ]; then
  TAG=$2
  for f in gpurun_out/${TAG}_cfg*_bench.json gpurun_out/${TAG}_cfg*_kernel_stats.csv; do [ -s "$f" ] && cp "$f" profiles/; done
  python tools/summarize_profiles.py $TAG
  exit 0
fi
TAG=${1:-r02}
mkdir -p gpurun_out
for C in 2 3 4pair 4lambda 5; do
  timeout 600 python bench.py --config $C > gpurun_out/${TAG}_cfg${C}_bench.json 2> gpurun_out/${TAG}_cfg${C}_bench.err; echo "bench cfg $C rc=$?"
done
timeout 600 python bench.py --config 5 --attention-dtype fp16 --no-cpu-baseline > gpurun_out/${TAG}_cfg5_fp16_bench.json 2> gpurun_out/${TAG}_cfg5_fp16_bench.err; echo "bench cfg5 fp16 rc=$?"
ULTR_FORCE_DP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-extras \
  > gpurun_out/${TAG}_cfg2_forced_dp_world1_bench.json 2> gpurun_out/${TAG}_cfg2_forced_dp_world1_bench.err; echo "forced dp rc=$?"
bash tools/profile_round.sh $TAG 2>&1 | grep "rc="
bash tools/profile_configs.sh $TAG 3 4pair 4lambda 5
timeout 300 python tools/fused_threshold.py 192 256 288 320 384 512 768 > gpurun_out/${TAG}_fused_threshold.txt 2>&1; echo "threshold rc=$?"
ULTR_BENCH_BIG=1 timeout 400 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_bench_configs_big.txt; echo "big rc=$?"
timeout 400 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_bench_configs.txt; echo "configs rc=$?"
