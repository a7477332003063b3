# GPU box: the parity suites under the non-default arithmetic / geometry knobs (every path that ships behind a knob stays green)
run() { echo "== $1 :: $2 ${3:+(-k \"$3\")}"; if [ -n "$3" ]; then env $1 python -m pytest $2 -x -q -m gpu -k "$3" 2>&1 | tail -1; else env $1 python -m pytest $2 -x -q -m gpu 2>&1 | tail -1; fi; }
run "ULTR_WG_H3=2" "tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_full_size.py tests/test_gpu_plugins.py tests/test_gpu_dp.py"
run "ULTR_WG_H3=0" "tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_setrank.py"
run "ULTR_FWD_R=32" "tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_full_size.py tests/test_gpu_plugins.py"
run "ULTR_BWD_R=32" "tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_full_size.py"
run "ULTR_BIG_FWD=2 ULTR_BIG_BWD=2" "tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_full_size.py"
run "ULTR_SR_ATTN_H3=0 ULTR_SR_WG_H3=0 ULTR_SR_H3=0" "tests/test_gpu_setrank.py" "not split_half"
run "ULTR_SR_ATTN_H3=2" "tests/test_gpu_setrank.py" "not full_size_properties"
run "ULTR_FB_H3=0 ULTR_FWD_H3=0 ULTR_BWD_H3=0 ULTR_WG_H3=0" "tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_pipeline.py"
run "ULTR_WGD=1" "tests/test_gpu_parity.py tests/test_gpu_edges.py"
run "ULTR_NO_FUSED_FB=1" "tests/test_gpu_parity.py tests/test_gpu_plugins.py tests/test_gpu_pipeline.py"
