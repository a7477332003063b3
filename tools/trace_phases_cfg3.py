"""GPU box: -DULTR_TRACE build, config 3's model through the separate forward / backward kernels: per-workgroup phase cycles.
(The weight-gradient launch stamps slots 8-12 as well: the forward's LN2 / sync columns are only meaningful where that launch has
no workgroup with the same index - the last lines.)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
src = sorted(os.path.join(ROOT, "ultra_pytorch_amd/csrc", f) for f in os.listdir(os.path.join(ROOT, "ultra_pytorch_amd/csrc")) if f.endswith(".hip"))
out = "/tmp/libultr_trace.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DULTR_TRACE"] + src + ["-o", out])
from ultra_pytorch_amd import _lib
lib = _lib.load(out)
_lib._LIB = lib
from ultra_pytorch_amd import hip_ops, engine, synthetic
from ultra_pytorch_amd.ranking_model import init_flat_params
F, L, B, H = 136, 20, 512, [512, 256, 128]
shape = hip_ops.DnnShape(F, H, "elu")
dev = torch.device("cuda")
p = init_flat_params(shape, 0).to(dev)
feats, ids, y = synthetic.make_batch(np.random.RandomState(0), B, L, F)
f, i_, y_ = torch.tensor(feats, device=dev), torch.tensor(ids, device=dev), torch.tensor(y, device=dev)
eng = engine.StepEngine(shape, B, L, dev)
ipw = torch.tensor(synthetic.load_ipw(), dtype=torch.float32, device=dev)
st = torch.zeros_like(p)
for _ in range(20):
    eng.train_step(p, st, f, feats.shape[0], i_, y_, ipw_table=ipw)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 32))()
lib.ultr_trace_read.argtypes = [ctypes.c_void_p]
lib.ultr_trace_read(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(64, 32)
nl = len(H) + 1
for blk in range(0, 20, 3):
    t = a[blk].astype(np.int64)
    s = ["gather+sync=%d" % (t[1] - t[0])]
    prev = t[1]
    for j in range(nl):
        s.append("LN%d=%d" % (j, t[2 + 3 * j] - prev))
        s.append("GEMM%d=%d" % (j, t[3 + 3 * j] - t[2 + 3 * j]))
        if j < nl - 1:
            s.append("sync=%d" % (t[4 + 3 * j] - t[3 + 3 * j]))
            prev = t[4 + 3 * j]
    print("fwd wg %4d:" % (blk * 32), " ".join(s), " total", t[3 + 3 * (nl - 1)] - t[0])
for blk in range(0, 20, 3):
    t = a[blk].astype(np.int64)
    s = ["loads+softmax=%d commit+sync=%d" % (t[14] - t[15], t[16] - t[14])]
    for jj in range(nl - 1):
        if jj > 0:
            s.append("fin+GEMM=%d sync=%d" % (t[17 + 4 * jj] - t[19 + 4 * (jj - 1)], t[18 + 4 * jj] - t[17 + 4 * jj]))
        s.append("rowpass%d=%d" % (nl - 1 - jj, t[19 + 4 * jj] - t[18 + 4 * jj]))
    print("bwd wg %4d:" % (blk * 32), " ".join(s), " total", t[19 + 4 * (nl - 2)] - t[15])
