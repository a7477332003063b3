#!/bin/bash
# GPU box: SetRank parity tests, config-5 step time, per-kernel rocprofv3 stats -> gpurun_out/srp
timeout 600 python -m pytest tests/test_gpu_setrank.py -x -q 2>&1 | tail -3
timeout 300 python tools/bench_configs.py 2>&1 | grep -E "^cfg5"
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/srp -o s -- python $R/tools/setrank_prof.py > $R/gpurun_out/srp.log 2>&1
