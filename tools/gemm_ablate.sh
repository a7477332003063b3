# GPU box: per-kernel averages of the split-half GEMM launches of config 5 for the product library and its timing variants
# (tools/ab_build.sh g_noa "-DUGEMM_H3_ABLATE=1" g_nostore "-DUGEMM_H3_ABLATE=2" g_nocross "-DUGEMM_H3_ABLATE=3")
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in product g_noa g_nostore g_nocross g_norescale g_noconv; do
if [ $v = product ]; then unset ULTR_HIP_LIB; else export ULTR_HIP_LIB=$REPO/ultra_pytorch_amd/lib/variants/libultr_$v.so; fi
rm -rf /tmp/ga_$v
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ga_$v -o s -- python $REPO/bench.py --config 5 --no-cpu-baseline --no-extras --steps 20 > /tmp/ga_$v.log 2>&1
python - <<PY
import csv,glob
fs=glob.glob('/tmp/ga_$v/**/*kernel_stats.csv',recursive=True)
print('$v')
if fs:
    for r in csv.DictReader(open(fs[0])):
        if 'gemm_h3' in r['Name']:
            n=r['Name']; tag=n[n.index('gemm_h3_kernelI')+15:n.index('gemm_h3_kernelI')+34]+' '+('EStore' if 'EStore' in n else 'EBiasRes' if 'EBiasRes' in n else 'EBiasAct')
            print('   %-40s calls %4s avg %7.1f us'%(tag, r['Calls'], float(r['AverageNs'])/1e3))
PY
done
