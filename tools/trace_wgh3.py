"""GPU box: -DULTR_TRACE build; phase cycles of dnn_wgrad_h3_kernel (wave 0 of every 32nd workgroup) at a BASELINE config.
   python tools/trace_wgh3.py [3|4]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ULTR_WG_H3"] = "1"
import numpy as np, torch
src = sorted(os.path.join(ROOT, "ultra_pytorch_amd/csrc", f) for f in os.listdir(os.path.join(ROOT, "ultra_pytorch_amd/csrc")) if f.endswith(".hip"))
out = "/tmp/libultr_trace.so"
extra = [a for a in sys.argv[2:]]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DULTR_TRACE", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"] + extra + src + ["-o", out],
                      stderr=subprocess.DEVNULL)
from ultra_pytorch_amd import _lib
lib = _lib.load(out)
_lib._LIB = lib
from ultra_pytorch_amd import hip_ops, engine, synthetic
from ultra_pytorch_amd.ranking_model import init_flat_params
cfg = sys.argv[1] if len(sys.argv) > 1 else "4"
F, L, B, H, algo = (136, 20, 512, [512, 256, 128], "dla") if cfg == "3" else (700, 50, 256, [512, 256, 128], "pairdebias")
shape = hip_ops.DnnShape(F, H, "elu")
dev = torch.device("cuda")
p = init_flat_params(shape, 0).to(dev)
feats, ids, y = synthetic.make_batch(np.random.RandomState(0), B, L, F)
f, i_, y_ = torch.tensor(feats, device=dev), torch.tensor(ids, device=dev), torch.tensor(y, device=dev)
eng = engine.StepEngine(shape, B, L, dev, algo=algo)
st = torch.zeros_like(p)
aux = torch.ones(2 * L if algo != "dla" else L + 1, device=dev)
for _ in range(10):
    eng.train_step(p, st if algo != "dla" else None, f, feats.shape[0], i_, y_, aux=aux)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (3 * 64 * 32))()
lib.ultr_trace_read.argtypes = [ctypes.c_void_p]
lib.ultr_trace_read(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(3, 64, 32).astype(np.int64)[0]
for blk in range(0, 17, 2):
    t = a[blk]
    if t[31] == 0:
        continue
    s = ["wg %3d tables=%d prologue=%d" % (32 * blk, t[1] - t[0], t[2] - t[1])]
    for k in (0, 2, 4):
        b = 2 + 4 * k
        if t[b + 3] == 0:
            break
        nxt = t[b + 8] if k < 4 else t[b + 3]
        s.append("| pub=%d mul+cv=%d pub=%d mul+cv=%d" % (t[b + 1] - t[b], t[b + 2] - t[b + 1], t[b + 3] - t[b + 2], nxt - t[b + 3]))
    s.append("|| loop=%d epi=%d" % (t[30] - t[1], t[31] - t[30]))
    print(" ".join(s))
