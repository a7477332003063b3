"""GPU box: shader clock / power while (a) the pure-MFMA probe, (b) the row-tile forward + backward of a big batch run back to back
(rocm-smi sampled from a side thread every 100 ms) - does the part hold its clock when the matrix cores and the L2 -> CU weight
stream are busy together?"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ultra_pytorch_amd import engine, hip_ops, synthetic
from ultra_pytorch_amd.ranking_model import init_flat_params
samples, stop = [], False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.perf_counter(), out))
        except Exception as e:
            samples.append((time.perf_counter(), "ERR %r" % e))
        time.sleep(0.1)
dev = torch.device("cuda")
F, hidden, B, L = 136, [512, 256, 128], 8192, 20
shape = hip_ops.DnnShape(F, hidden, "elu")
eng = engine.StepEngine(shape, B, L, dev, algo="dla", learning_rate=0.05)
p = init_flat_params(shape, 0).to(dev)
f, i, y = synthetic.make_batch(np.random.RandomState(0), B, L, F)
f, nd, i, y = torch.tensor(f, device=dev), f.shape[0], torch.tensor(i, device=dev), torch.tensor(y, device=dev)
aux = torch.zeros(L + 1, device=dev)
for _ in range(5): eng.train_step(p, None, f, nd, i, y, aux=aux)
torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
time.sleep(0.5)
t0 = time.perf_counter()
a = torch.randn(8192, 8192, device=dev); b = torch.randn(8192, 8192, device=dev)
while time.perf_counter() - t0 < 3.0:
    for _ in range(10): a @ b
    torch.cuda.synchronize()
t1 = time.perf_counter()
while time.perf_counter() - t1 < 4.0:
    for _ in range(50): eng.train_step(p, None, f, nd, i, y, aux=aux)
    torch.cuda.synchronize()
t2 = time.perf_counter()
time.sleep(0.5)
stop = True; th.join()
import json, re
def fields(txt):
    try:
        d = json.loads(txt); c = d[list(d)[0]]
        return {k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower() or "fclk" in k.lower() or "mclk" in k.lower()}
    except Exception:
        return txt[:200]
for t, txt in samples[::3]:
    phase = "idle" if t < t0 else "fp32 GEMM (torch / hipBLASLt)" if t < t1 else "training steps (163 840 rows)" if t < t2 else "idle"
    print("%6.2f s  %-32s %s" % (t - t0, phase, fields(txt)))
