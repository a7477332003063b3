#!/bin/bash
# registers / spills of the kernels of one translation unit:  tools/kernel_resources.sh ultr_sr_bwd.hip [extra hipcc flags]
cd "$(dirname "$0")/../ultra_pytorch_amd/csrc"
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Xclang -target-feature -Xclang -packed-fp32-ops \
  -Rpass-analysis=kernel-resource-usage "$@" -c $src -o /tmp/kres.o 2>&1 | grep "error\|Function Name\| VGPRs:\|VGPRs Spill\|LDS Size" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//'
