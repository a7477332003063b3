"""GPU box: phase times of dnn_fwdw_kernel (wide-tile forward) from a -DULTR_TRACE build (tools/ab_build.sh trace "-DULTR_TRACE"):
   ULTR_TRACE_LIB=ultra_pytorch_amd/lib/variants/libultr_trace.so python tools/trace_fwdw.py [3|4]
Stamps (wave 0 of every 32nd workgroup, s_memtime = 100 MHz ticks): 0 start, 1 prologue done, per layer j: 2+3j LayerNorm done,
3+3j barrier passed, 4+3j product + epilogue done."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ultra_pytorch_amd import _lib
lib = _lib.load(os.environ["ULTR_TRACE_LIB"])
_lib._LIB = lib
from ultra_pytorch_amd import hip_ops, engine, synthetic
from ultra_pytorch_amd.ranking_model import init_flat_params
cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
F, L, B, H = {"3": (136, 20, 512, [512, 256, 128]), "4": (700, 50, 256, [512, 256, 128])}[cfg]
shape = hip_ops.DnnShape(F, H, "elu")
dev = torch.device("cuda")
p = init_flat_params(shape, 0).to(dev)
feats, ids, y = synthetic.make_batch(np.random.RandomState(0), B, L, F)
f, i_ = torch.tensor(feats, device=dev), torch.tensor(ids, device=dev)
eng = engine.StepEngine(shape, B, L, dev)
print("tile rows:", lib.ultr_dnn_forward_tile_rows(shape.desc, B * L, 1))
for _ in range(20):
    eng.forward(p, f, feats.shape[0], i_, train=True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (3 * 64 * 32))()
lib.ultr_trace_read.argtypes = [ctypes.c_void_p]
lib.ultr_trace_read(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(3, 64, 32).astype(np.int64)[1]  # bank 1
nl = len(H) + 1
t0 = a[:8, 30].min()
print("100 MHz counter: start / end of the sampled workgroups relative to the first start (us):",
      " ".join("%.1f/%.1f" % ((a[b, 30] - t0) / 100.0, (a[b, 31] - t0) / 100.0) for b in range(8)))
for blk in range(0, 8):
    t = a[blk]
    out = ["prologue=%d" % (t[1] - t[0])]
    for j in range(nl):
        prev = t[1] if j == 0 else t[4 + 3 * (j - 1)]
        out.append("LN%d=%d" % (j, t[2 + 3 * j] - prev))
        if j < nl - 1:
            out.append("sync=%d product%d=%d" % (t[3 + 3 * j] - t[2 + 3 * j], j, t[4 + 3 * j] - t[3 + 3 * j]))
    print("wg %4d:" % (blk * 32), " ".join(out), " total=%d ticks (x10 ns)" % (t[2 + 3 * (nl - 1)] - t[0]))
