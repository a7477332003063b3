"""GPU box: config 5 forward scores with the split-half attention vs the fp32 matrix-core attention - where do they differ?
   ULTR_SR_ATTN_H3=0 python tools/sr_attn_diff.py save ; ULTR_SR_ATTN_H3=1 python tools/sr_attn_diff.py cmp"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.test_gpu_setrank import run_step
from ultra_pytorch_amd import hip_ops, synthetic
from ultra_pytorch_amd.ranking_model.SetRank import init_setrank_params
F, dm, H, nl, dff, B, L = 220, 256, 8, int(os.environ.get("NL", "2")), 64, 1024, 100
shape = hip_ops.SetRankShape(F, dm, H, nl, dff)
rng = np.random.RandomState(5)
feats, ids, y = synthetic.make_batch(rng, B, L, F)
p0 = init_setrank_params(shape, seed=3).numpy()
kw = dict(learning_rate=0.05, max_gradient_norm=5.0)
s1, g1, _, _, sc1 = run_step(shape, B, L, kw, p0, np.zeros_like(p0), feats, ids, y, None)
if sys.argv[1] == "save":
    np.save("/tmp/sr_s_ref.npy", s1); np.save("/tmp/sr_g_ref.npy", g1)
else:
    r = np.load("/tmp/sr_s_ref.npy"); gr = np.load("/tmp/sr_g_ref.npy")
    dd = np.abs(s1 - r)
    print("scores: max %.3e 99.99%% %.3e 99%% %.3e" % (dd.max(), np.quantile(dd, 0.9999), np.quantile(dd, 0.99)))
    idx = np.argsort(dd.ravel())[::-1][:12]
    for k in idx:
        b, l = divmod(int(k), L)
        print("  list %4d pos %3d diff %.3e score %.5f  list-max diff %.3e" % (b, l, dd[b, l], r[b, l], dd[b].max()))
    per_list = dd.max(axis=1)
    print("lists with max diff > 5e-6: %d of %d;  by position (mean diff): first16 %.2e last4(96..99) %.2e" % ((per_list > 5e-6).sum(), B, dd[:, :16].mean(), dd[:, 96:].mean()))
    n = shape.n_params
    gd = np.abs(g1[:n] - gr[:n])
    print("grads: max diff %.3e  max|g| %.3e" % (gd.max(), np.abs(gr[:n]).max()))
