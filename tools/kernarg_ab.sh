for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v"
  for rep in 1 2; do
    HIP_FORCE_DEV_KERNARG=$v ULTR_HIP_LIB=$PWD/ultra_pytorch_amd/lib/variants/libultr_fbwt.so timeout 120 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(1e3*d['ms_per_step'],2), d['kernel_us'])"
  done
done
