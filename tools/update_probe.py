"""GPU box: ultr_apply_update alone at a config's parameter count - with the weight copies to maintain (update_tiled_kernel) and without
(update_kernel: one element per thread, nothing but params / state / grads).  The difference is what the copies cost.
   python tools/update_probe.py [config]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ultra_pytorch_amd import _lib, hip_ops  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "4pair"
W = bench.Workload(key, torch.device("cuda:0"))
for i in range(20):
    W.step(i)
torch.cuda.synchronize()
eng, lib = W.eng, _lib.load()
wt = hip_ops.weight_copy(W.shape).get(W.params)


def run(with_copies, n=300):
    args = (ctypes.byref(eng.udesc), ctypes.byref(W.shape.desc), W.params.data_ptr(), wt.data_ptr() if with_copies else None,
            W.state.data_ptr() if W.state is not None else None, eng.grads.data_ptr(), W.aux.data_ptr() if W.aux is not None else None,
            eng.bwd_ws.data_ptr(), eng.scalars.data_ptr(), hip_ops.raw_stream())
    for _ in range(20):
        lib.ultr_apply_update(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        lib.ultr_apply_update(*args)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


eng.grads.mul_(1e-6)  # repeated updates must not walk the weights out of the split-half range
for rep in range(2):
    print("config %s (%d parameters): update with the weight copies %.2f us, without %.2f us (back-to-back launches)" %
          (key, W.P, run(True), run(False)), flush=True)
