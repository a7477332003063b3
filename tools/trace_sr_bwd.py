"""GPU box: phase cycles of sr_bwd_ffn_kernel / sr_bwd_proj_kernel (ultr_sr_bwd.hip, config 5's shape) from a -DULTR_TRACE build:
   tools/ab_build.sh trace "-DULTR_TRACE"; ULTR_TRACE_LIB=ultra_pytorch_amd/lib/variants/libultr_trace.so python tools/trace_sr_bwd.py
Wave 0 of workgroups 0, 8, .. 56 on their SECOND tile (steady state: the tile's rows were requested during the first)."""
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
eng = engine.SetRankStepEngine(shape, B, L, dev, algo="softmax", learning_rate=0.05, max_gradient_norm=5.0)
p = init_setrank_params(shape, seed=3).to(dev)
f, i_, yy = torch.tensor(feats, device=dev), torch.tensor(ids, device=dev), torch.tensor(y, device=dev)
ipw = torch.tensor(np.asarray(synthetic.load_ipw(), np.float32), device=dev)
for _ in range(3):
    eng.forward(p, f, feats.shape[0], i_, train=True)
    eng.loss(yy, ipw_table=ipw)
    eng.backward(p, f, feats.shape[0], i_)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 32))()
lib.ultr_srb_trace_read.argtypes = [ctypes.c_void_p]
lib.ultr_srb_trace_read(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(64, 32).astype(np.int64)
ffn = ["rows(LN2')", "sync", "product dF", "request", "wgrad Wf2", "sync", "raw+mask+planes", "product dOut1", "sync"]
proj = ["rows(LN1')", "sync", "product dA", "request", "wgrad Wf1", "sync"]
for nm, names, base in (("sr_bwd_ffn_kernel", ffn, 0), ("sr_bwd_proj_kernel", proj, 16)):
    print(nm, "(s_memtime ticks = shader clocks, ~2.4 GHz)")
    for blk in range(0, 64, 8):
        t = a[blk, base:base + len(names) + 1]
        print("  wg %3d:" % blk, " ".join("%s=%d" % (names[k], t[k + 1] - t[k]) for k in range(len(names))), " tile=%d" % (t[len(names)] - t[0]))

if hasattr(lib, "ultr_srf_trace_read"):
    lib.ultr_srf_trace_read.argtypes = [ctypes.c_void_p]
    lib.ultr_srf_trace_read(buf)
    a = np.array(buf[:], dtype=np.uint64).reshape(64, 32).astype(np.int64)
    names = ["rows(A, x)", "product s1 (+sync)", "LN1 (+sync)", "f (+sync)", "product s2", "request+sync", "LN2", "head", "sync"]
    print("sr_fwd_block_kernel (the LAST block's launch: with the output FFN)")
    for blk in range(0, 64, 8):
        t = a[blk, :10]
        print("  wg %3d:" % blk, " ".join("%s=%d" % (names[k], t[k + 1] - t[k]) for k in range(9)), " tile=%d" % (t[9] - t[0]))
