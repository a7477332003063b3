"""GPU box: where do the microseconds of ONE synchronous training step go on the HOST side (config 2)?
  enqueue : host time of StepEngine.train_step() alone (the GPU queue absorbs the launches: 64 steps behind one sync)
  python  : the same with the C call replaced by a no-op (pure Python / ctypes marshalling)
  synced  : step + read_loss() per step (what `value` times) and step + loss.item()-style stream sync + D2H
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ultra_pytorch_amd import engine, hip_ops, synthetic  # noqa: E402
from ultra_pytorch_amd.ranking_model import init_flat_params  # noqa: E402

F, L, B, H = 136, 10, 256, [256, 256]
dev = torch.device("cuda", 0)
shape = hip_ops.DnnShape(F, H, "elu")
p = init_flat_params(shape, seed=0).to(dev)
st = torch.zeros_like(p)
ipw = torch.tensor(synthetic.load_ipw(), dtype=torch.float32, device=dev)
rng = np.random.RandomState(0)
pool = []
for _ in range(16):
    f, i, y = synthetic.make_batch(rng, B, L, F)
    pool.append((torch.from_numpy(f).to(dev), f.shape[0], torch.from_numpy(i).to(dev), torch.from_numpy(y).to(dev)))
eng = engine.StepEngine(shape, B, L, dev, algo="softmax")


def step(k):
    f, nd, ids, y = pool[k % 16]
    return eng.train_step(p, st, f, nd, ids, y, ipw_table=ipw)


for k in range(200):
    step(k)
torch.cuda.synchronize()
res = {}
# enqueue only
ts = []
for rep in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(64):
        step(k)
    ts.append((time.perf_counter() - t0) / 64)
    torch.cuda.synchronize()
res["enqueue_us"] = 1e6 * float(np.median(ts))
# python only: swap the C entry point for a no-op of the same signature
real = eng._fn
NOOP = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p)(lambda a, s: 0)
eng._fn = NOOP
ts = []
for rep in range(20):
    t0 = time.perf_counter()
    for k in range(64):
        step(k)
    ts.append((time.perf_counter() - t0) / 64)
res["python_only_us (incl. a ctypes callback round trip)"] = 1e6 * float(np.median(ts))
eng._fn = real
step(0)
eng.read_loss()
# synced variants
for name, reader in (("read_loss", lambda sc: eng.read_loss()), ("item", lambda sc: sc[0].item())):
    for k in range(50):
        reader(step(k))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1000):
        reader(step(k))
    res["synced_%s_us" % name] = 1e6 * (time.perf_counter() - t0) / 1000
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(2000):
    step(k)
torch.cuda.synchronize()
res["nosync_us"] = 1e6 * (time.perf_counter() - t0) / 2000
# latency of the report alone: a lone update launch -> host sees the sequence number
eng.udesc.seq = 12345
ts = []
for rep in range(200):
    eng._next_seq()
    t0 = time.perf_counter()
    eng.update(p, st, None)
    eng.read_scalars()
    ts.append(time.perf_counter() - t0)
res["lone_update_launch_to_report_us"] = 1e6 * float(np.median(ts))
print(res)
