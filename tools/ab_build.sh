#!/bin/bash
# Build experiment variants of the library next to the product one:  tools/ab_build.sh name "-DFOO=1" [name2 "-DBAR=2" ...]
# -> ultra_pytorch_amd/lib/variants/libultr_<name>.so ; run with ULTR_HIP_LIB=<path> python bench.py ...
cd "$(dirname "$0")/.."
mkdir -p ultra_pytorch_amd/lib/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -Xclang -target-feature -Xclang -packed-fp32-ops $flags ultra_pytorch_amd/csrc/*.hip \
    -o ultra_pytorch_amd/lib/variants/libultr_$name.so &
done
wait
ls -la ultra_pytorch_amd/lib/variants/
