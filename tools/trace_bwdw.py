"""GPU box: phase times of dnn_bwdw_kernel (wide-tile row-local backward) from a -DULTR_TRACE build:
   ULTR_TRACE_LIB=ultra_pytorch_amd/lib/variants/libultr_trace.so python tools/trace_bwdw.py [3|4]
Stamps (wave 0 of every 32nd workgroup, shader clock): 0 start, per layer j = top .. 1 with jj = top - j: 1+3jj row pass done
(+ the barrier in it), 2+3jj column partials folded, 3+3jj product done.  30 / 31: start / end on the shared 100 MHz counter."""
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
f, i_, y_ = torch.tensor(feats, device=dev), torch.tensor(ids, device=dev), torch.tensor(y, device=dev)
eng = engine.StepEngine(shape, B, L, dev, algo="dla")
aux = torch.zeros(L + 1, device=dev)
print("backward tile rows:", lib.ultr_dnn_backward_tile_rows(shape.desc, B * L))
for _ in range(10):
    eng.train_step(p, None, f, feats.shape[0], i_, y_, aux=aux)
    eng.read_loss()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (3 * 64 * 32))()
lib.ultr_trace_read.argtypes = [ctypes.c_void_p]
lib.ultr_trace_read(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(3, 64, 32).astype(np.int64)[2]  # bank 2
top = len(H)
t0 = a[:8, 30].min()
print("100 MHz counter: start / end of the sampled workgroups relative to the first start (us):",
      " ".join("%.1f/%.1f" % ((a[b, 30] - t0) / 100.0, (a[b, 31] - t0) / 100.0) for b in range(8)))
for blk in range(0, 8):
    t = a[blk]
    out, prev = [], t[0]
    for jj in range(top):
        j = top - jj
        out.append("rowpass%d=%d fold=%d" % (j, t[1 + 3 * jj] - prev, t[2 + 3 * jj] - t[1 + 3 * jj]))
        prev = t[2 + 3 * jj]
        if j > 1:
            out.append("product%d=%d" % (j - 1, t[3 + 3 * jj] - prev))
            prev = t[3 + 3 * jj]
    print("wg %4d:" % (blk * 32), " ".join(out), " total=%d cycles" % (prev - t[0]))
    print("         row pass 1: issue loads=%d first pass (waits for x)=%d wave sums=%d second pass + planes=%d barrier=%d partials=%d"
          % (t[12] - t[3 + 3 * (top - 2)], t[13] - t[12], t[14] - t[13], t[15] - t[14], t[16] - t[15], t[1 + 3 * (top - 1)] - t[16]))
