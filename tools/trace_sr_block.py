"""GPU box: phase cycles of sr_block_fwd_kernel (SetRank's fused encoder block, config 5's shape) from a -DULTR_TRACE build:
   ULTR_TRACE_LIB=ultra_pytorch_amd/lib/variants/libultr_trace.so python tools/trace_sr_block.py
Wave 0 of every 8th workgroup (the first 64 of them); the forward's LAST block launch wins (the slots are shared with the GEMM kernels'
stamps: 20 start, 21 prologue done, 22 product Wd done, 23 barrier, 24 LayerNorm 1 done, 25 barrier, 26 f done, 27 barrier, 28 product
Wf2 done, 29 barrier, 30 LayerNorm 2 done)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ultra_pytorch_amd import _lib
lib = _lib.load(os.environ["ULTR_TRACE_LIB"])
_lib._LIB = lib
from ultra_pytorch_amd import hip_ops, synthetic, engine
from ultra_pytorch_amd.ranking_model.SetRank import init_setrank_params
F, dm, H, nl, dff, B, L = 220, 256, 8, 2, 64, 1024, 100
shape = hip_ops.SetRankShape(F, dm, H, nl, dff)
rng = np.random.RandomState(5)
feats, ids, y = synthetic.make_batch(rng, B, L, F)
dev = torch.device("cuda")
p = init_setrank_params(shape, seed=3).to(dev)
saved = torch.zeros(shape.saved_bytes(B * L) // 4, device=dev)
scores = torch.zeros(B, L, device=dev)
f, i_ = torch.tensor(feats, device=dev), torch.tensor(ids, device=dev)
lib.ultr_gemm_trace_arm(1)
for _ in range(3):
    hip_ops.setrank_forward(shape, p, f, feats.shape[0], i_, B, L, scores, saved)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 32))()
lib.ultr_gemm_trace_read.argtypes = [ctypes.c_void_p]
lib.ultr_gemm_trace_read(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(64, 32).astype(np.int64)
names = ["prologue", "product Wd", "sync", "LN1", "sync", "f", "sync", "product Wf2", "sync", "LN2"]
for blk in range(0, 64, 8):
    t = a[blk]
    print("wg %4d:" % (8 * blk), " ".join("%s=%d" % (names[k], t[21 + k] - t[20 + k]) for k in range(10)), " total=%d" % (t[30] - t[20]))
