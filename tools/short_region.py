"""GPU box: what a K = 20 step timed region (the driver's bench call) pays on top of the steady-state step."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ultra_pytorch_amd import engine, hip_ops, synthetic, _lib
from ultra_pytorch_amd.ranking_model import init_flat_params
dev = torch.device("cuda")
F, hidden, B, L = 136, [256, 256], 256, 10
shape = hip_ops.DnnShape(F, hidden, "elu")
eng = engine.StepEngine(shape, B, L, dev, algo="softmax", learning_rate=0.05)
p = init_flat_params(shape, 0).to(dev); st = torch.zeros_like(p)
f, i, y = synthetic.make_batch(np.random.RandomState(0), B, L, F)
f, nd, i, y = torch.tensor(f, device=dev), f.shape[0], torch.tensor(i, device=dev), torch.tensor(y, device=dev)
ipw = torch.tensor(synthetic.load_ipw(), dtype=torch.float32, device=dev)
step = lambda: eng.train_step(p, st, f, nd, i, y, ipw_table=ipw)
for _ in range(200): step()
torch.cuda.synchronize()
lib = _lib.load()
def region(K, reps=30):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K): step()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6)
    ts.sort(); return ts[len(ts) // 2], ts[0]
for K in (2000, 200, 20):
    print("K=%4d plain: median %.2f us/step, best %.2f" % ((K,) + region(K, 10 if K > 200 else 30)))
lib.ultr_prof_set_stride(8); lib.ultr_prof_enable(0xBF, 7 * 4000)
print("K=  20 with the kernel timers armed (stride 8): median %.2f us/step, best %.2f" % region(20))
tot, cnt = (ctypes.c_double * 8)(), (ctypes.c_int64 * 8)(); lib.ultr_prof_collect(tot, cnt); lib.ultr_prof_enable(0, 0)
t0 = time.perf_counter()
for _ in range(2000): step()
print("host time per train_step call (GPU-bound loop excluded: issue only) %.2f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
torch.cuda.synchronize()
# accuracy of timing ONE kernel (the fused forward+loss+backward, slot 7) against timing all of them
for mask, name in ((0xBF, "all kernels timed"), (1 << 7, "only the fused kernel timed")):
    for stride in (8, 32):
        lib.ultr_prof_set_stride(stride); lib.ultr_prof_enable(mask, 7 * 4000)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2000): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2000 * 1e6
        lib.ultr_prof_collect(tot, cnt); lib.ultr_prof_enable(0, 0)
        print("%-28s stride %2d: %.2f us/step; fused kernel %.2f us over %d samples" % (name, stride, dt, 1e3 * tot[7] / max(cnt[7], 1), cnt[7]))
lib.ultr_prof_set_stride(1)
