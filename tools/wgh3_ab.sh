run() { echo -n "$1 | $2: "; env $1 timeout 300 python bench.py --config $2 --no-cpu-baseline --no-extras --steps 300 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernel_us'); print(round(1e3*d['ms_per_step'],2), 'wgrad', k.get('dnn_wgrad_kernel'), 'reduce', k.get('grad_reduce_kernel'))"; }
V=$PWD/ultra_pytorch_amd/lib/variants
for c in ${CFGS:-4pair}; do
for v in "$@"; do
if [ $v = product ]; then run "ULTR_WG_H3=1 ULTR_WG_H3_WGS=${WGS:-544}" $c; else run "ULTR_WG_H3=1 ULTR_WG_H3_WGS=${WGS:-544} ULTR_HIP_LIB=$V/libultr_$v.so" $c; fi
done
done
