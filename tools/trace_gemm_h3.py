"""GPU box: -DULTR_TRACE build; phase cycles of ugemm::gemm_h3_kernel (wave 0 of every 8th workgroup) for ONE launch: SetRank's
attention-output projection shape (102 400 x 256 -> 256 with residual) driven through the model's forward; the LAST GEMM launch of the
forward that stamped wins (slots are overwritten), so this reads the output layer's... use with care: prints the first chunk's phases."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
src = sorted(os.path.join(ROOT, "ultra_pytorch_amd/csrc", f) for f in os.listdir(os.path.join(ROOT, "ultra_pytorch_amd/csrc")) if f.endswith(".hip"))
out = "/tmp/libultr_trace.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DULTR_TRACE", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"] + src + ["-o", out], stderr=subprocess.DEVNULL)
from ultra_pytorch_amd import _lib
lib = _lib.load(out)
_lib._LIB = lib
from tests.test_gpu_setrank import run_step
from ultra_pytorch_amd import hip_ops, synthetic
from ultra_pytorch_amd.ranking_model.SetRank import init_setrank_params
F, dm, H, nl, dff, B, L = 220, 256, 8, 2, 64, 1024, 100
shape = hip_ops.SetRankShape(F, dm, H, nl, dff)
rng = np.random.RandomState(5)
feats, ids, y = synthetic.make_batch(rng, B, L, F)
p0 = init_setrank_params(shape, seed=3).numpy()
run_step(shape, B, L, dict(learning_rate=0.05, max_gradient_norm=5.0), p0, np.zeros_like(p0), feats, ids, y, None)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 32))()
lib.ultr_gemm_trace_read.argtypes = [ctypes.c_void_p]
lib.ultr_gemm_trace_read(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(64, 32).astype(np.int64)
for blk in range(0, 64, 6):
    t = a[blk]
    if t[21] == 0:
        continue
    s = ["wg %3d first=%d" % (8 * blk, t[1] - t[0])]
    for k in (0, 2):
        b = 2 + 3 * k
        s.append("| mul=%d st+ld=%d bar=%d mul=%d st+ld=%d bar=%d" % (t[b + 1] - t[b], t[b + 2] - t[b + 1], t[b + 3] - t[b + 2], t[b + 4] - t[b + 3], 0, t[b + 5] - t[b + 4]))
    s.append("|| loop=%d epilogue=%d" % (t[20] - t[1], t[21] - t[20]))
    print(" ".join(s))
