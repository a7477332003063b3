for t in 224 448 476 504; do
  echo "== WGRAD_WGS=$t"
  ULTR_WGRAD_WGS=$t timeout 120 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(1e3*d['ms_per_step'],2), d['kernel_us'])"
done
