"""GPU box: would a captured HIP graph shorten the step?  Config 2 (and any other: argv[1]) - the step's launches (ultr_train_step)
captured once through torch.cuda.CUDAGraph and replayed, against the same launches issued directly, no host synchronisation inside
either loop.  A TIMING probe: a replay bakes the step's sequence number and batch pointers, so its numbers mean nothing.
   python tools/graph_probe.py [config] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
dev = torch.device("cuda:0")
W = bench.Workload(key, dev)
for i in range(200):
    W.step(i)
torch.cuda.synchronize()


def direct(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        W.step(0)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3):
        W.step(0)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g, stream=s):
        W.step(0)
except Exception as e:  # a call the capture refuses
    print("capture failed:", repr(e)[:300])
    sys.exit(0)


def replay(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        g.replay()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for rep in range(3):
    print("config %s  direct %.2f us/step   graph replay %.2f us/step" % (key, direct(n), replay(n)), flush=True)
