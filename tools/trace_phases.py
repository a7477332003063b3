"""GPU box: build a -DULTR_TRACE copy of the library, run the forward kernel at the bench shape, print phase deltas."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
src = sorted(os.path.join(ROOT, "ultra_pytorch_amd/csrc", f) for f in os.listdir(os.path.join(ROOT, "ultra_pytorch_amd/csrc")) if f.endswith(".hip"))
out = os.environ.get("ULTR_TRACE_LIB", "/tmp/libultr_trace.so")  # a variant prebuilt with tools/ab_build.sh trace "-DULTR_TRACE", or built here
if not os.path.exists(out):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-Xclang", "-target-feature",
                           "-Xclang", "-packed-fp32-ops", "-DULTR_TRACE"] + src + ["-o", out])
from ultra_pytorch_amd import _lib
lib = _lib.load(out)
_lib._LIB = lib
from ultra_pytorch_amd import hip_ops, engine, synthetic
from ultra_pytorch_amd.ranking_model import init_flat_params
F, L, B, H = 136, 10, 256, [256, 256]
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
buf = (ctypes.c_ulonglong * (3 * 64 * 32))()
lib.ultr_trace_read.argtypes = [ctypes.c_void_p]
lib.ultr_trace_read(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(3, 64, 32)[0]  # bank 0: the 8-wave kernels
# fused forward+loss+backward kernel (dnn_fb_kernel): 0 start, 1 prologue done, 2+2j LayerNorm_j done, 3+2j GEMM_j done,
# 16 loss done, then the backward stamps of dnn_bwd2_kernel (18+4jj row pass start, 19+4jj row pass done, 17+4jj GEMM done)
if os.environ.get("ULTR_NO_FUSED_FB", "0") != "1":
    for blk in range(3):
        t = a[blk].astype(np.int64)
        print("fused wg %3d: prologue=%d LN0=%d GEMM0=%d LN1=%d GEMM1=%d LN2+score=%d loss=%d | rowcol2=%d fin+GEMM1'=%d sync=%d rowcol1=%d "
              "fin+GEMM0'=%d sync=%d rowcol0=%d  total(0->27)=%d" % (blk * 32, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4],
                                                               t[6] - t[5], t[16] - t[6], t[19] - t[18], t[21] - t[19], t[22] - t[21],
                                                               t[23] - t[22], t[25] - t[23], t[26] - t[25], t[27] - t[26], t[27] - t[0]))
    for blk in range(3):
        t = a[blk].astype(np.int64)
        print("fused wg %3d detail: prologue: start->params in LDS=%d ->features in LDS=%d ->barrier=%d | LN0: ->row sums=%d ->var=%d "
              "->written=%d ->barrier=%d | rowcol1 done->kernel end=%d  total(0->13)=%d"
              % (blk * 32, t[28] - t[0], t[29] - t[28], t[1] - t[29], t[30] - t[1], t[31] - t[30], t[7] - t[31], t[2] - t[7], t[13] - t[23], t[13] - t[0]))
    for blk in range(0, 7, 1):
        t = a[blk].astype(np.int64)
        print("wgrad wg %3d: preamble=%d mainloop=%d ldswrite+sync=%d reduce+store=%d total=%d" % (blk * 32, t[9] - t[8], t[10] - t[9], t[11] - t[10], t[12] - t[11], t[12] - t[8]))
    sys.exit(0)
names = ["start", "gather+sync", "LN0", "GEMM0", "sync", "LN1", "GEMM1", "sync", "LN2", "dot"]
for blk in range(5):
    t = a[blk].astype(np.int64)
    print("wg %3d:" % (blk * 32), " ".join("%s=%d" % (names[k], t[k] - t[k - 1]) for k in range(1, 10)), " total", t[9] - t[0])

for blk in range(3):
    t = a[blk].astype(np.int64)
    print("fwd LN1 detail wg %3d: postGEMMbarrier->LNstart=%d loads+mean=%d var=%d write=%d ->barrier=%d" % (blk * 32, t[28] - t[4], t[29] - t[28], t[30] - t[29], t[31] - t[30], t[5] - t[31]))
# fast backward kernel (dnn_bwd2_kernel): 15 start, 14 softmax done, 16 prologue done; per layer jj = top - j:
# 17+4jj GEMM done, 18+4jj barrier passed (18 for the top layer = right after 16), 19+4jj row pass done
for blk in range(3):
    t = a[blk].astype(np.int64)
    out = ["loads+softmax=%d commit+sync=%d" % (t[14] - t[15], t[16] - t[14])]
    out.append("L2: rowcol=%d commit+sync+finalize+gemm=%d" % (t[19] - t[18], t[21] - t[19]))
    out.append("L1: sync=%d rowcol=%d | next gemm(incl commit,sync,finalize)=%d" % (t[22] - t[21], t[23] - t[22], t[25] - t[23]))
    out.append("L0: sync=%d rowcol=%d" % (t[26] - t[25], t[27] - t[26]))
    print("bwd2 wg %3d:" % (blk * 32), " | ".join(out), " total(15->27)", t[27] - t[15])
for blk in range(0, 13, 3):
    t = a[blk].astype(np.int64)
    print("wgrad wg %3d: ids+sync=%d mainloop=%d ldswrite+sync=%d reduce+store=%d total=%d" % (blk * 32, t[9] - t[8], t[10] - t[9], t[11] - t[10], t[12] - t[11], t[12] - t[8]))
