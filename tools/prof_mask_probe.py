"""GPU box: what the fused kernel's dispatch-timestamp duration comes out as, depending on WHICH kernels of the step carry timers:
all of them, the fused kernel alone, the fused kernel + the update kernel in front of it."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from ultra_pytorch_amd import _lib, engine, hip_ops, synthetic
from ultra_pytorch_amd.ranking_model import init_flat_params
F, hidden, B, L = 136, [256, 256], 256, 10
shape = hip_ops.DnnShape(F, hidden, "elu")
lib = shape.lib
feats, ids, y = synthetic.make_batch(np.random.RandomState(5), B, L, F)
ipw = np.asarray(synthetic.load_ipw(), np.float32)
dev = lambda a, dt=torch.float32: torch.as_tensor(a).to("cuda", dt)
p0 = init_flat_params(shape, seed=3).numpy()
eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax", learning_rate=0.05, max_gradient_norm=5.0)
params, state = dev(p0.copy()), dev(np.zeros_like(p0))
f, i, yy, tab = dev(feats), dev(ids, torch.int32), dev(y), dev(ipw)
step = lambda: eng.train_step(params, state, f, feats.shape[0], i, yy, ipw_table=tab)
for _ in range(300): step()
torch.cuda.synchronize()
tot, cnt = (ctypes.c_double * 8)(), (ctypes.c_int64 * 8)()
for name, mask in (("all", 0xBF), ("fused only", 1 << 7), ("fused + update", (1 << 7) | (1 << 5)), ("fused + reduce + update", (1 << 7) | (1 << 5) | (1 << 4)), ("all", 0xBF), ("fused only", 1 << 7)):
    for sync in (False, True):
        lib.ultr_prof_set_stride(1)
        _lib.check(lib.ultr_prof_enable(mask, 8 * 200), "enable")
        for _ in range(200):
            step()
            if sync: eng.read_loss()
        torch.cuda.synchronize()
        _lib.check(lib.ultr_prof_collect(tot, cnt), "collect")
        lib.ultr_prof_enable(0, 0)
        print("%-26s %-22s fused %.2f us  (wgrad %.2f reduce %.2f update %.2f)" % (name, "host reads every step" if sync else "back to back", 1e3 * tot[7] / max(cnt[7], 1),
              1e3 * tot[3] / max(cnt[3], 1), 1e3 * tot[4] / max(cnt[4], 1), 1e3 * tot[5] / max(cnt[5], 1)))
