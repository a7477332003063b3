"""GPU box: SetRank config 5 at full size - how far do the scores move when the documents of every list are permuted (they must
not: the encoder is permutation-equivariant)?  Run once per attention plan:  ULTR_SR_ATTN_H3=0|1 python tools/sr_perm_diag.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.test_gpu_setrank import run_step
from ultra_pytorch_amd import hip_ops, synthetic
from ultra_pytorch_amd.ranking_model.SetRank import init_setrank_params
from oracle import ultr_oracle as O
F, dm, H, nl, dff, B, L = 220, 256, 8, 2, 64, 1024, 100
shape = hip_ops.SetRankShape(F, dm, H, nl, dff)
rng = np.random.RandomState(5)
feats, ids, y = synthetic.make_batch(rng, B, L, F)
p0 = init_setrank_params(shape, seed=3).numpy()
kw = dict(learning_rate=0.05, max_gradient_norm=5.0)
s1, g1, _, _, sc1 = run_step(shape, B, L, kw, p0, np.zeros_like(p0), feats, ids, y, None)
perm = rng.permutation(L)
s2, g2, _, _, sc2 = run_step(shape, B, L, kw, p0, np.zeros_like(p0), feats, ids[perm], y[perm], None)
dd = np.abs(s2 - s1[:, perm])
print("H3 attention:", os.environ.get("ULTR_SR_ATTN_H3", "1"), " perm diff: max %.3e  99.99%% %.3e  99%% %.3e  mean %.3e" % (dd.max(), np.quantile(dd, 0.9999), np.quantile(dd, 0.99), dd.mean()))
nref = 24
ref = O.setrank_forward(torch.from_numpy(p0), F, dm, H, nl, dff, feats, ids[:, :nref]).detach().numpy()
de = np.abs(s1[:nref] - ref)
print("  vs oracle on %d lists: max %.3e  99%% %.3e  mean %.3e" % (nref, de.max(), np.quantile(de, 0.99), de.mean()))
n = shape.n_params
gd = np.abs(g2[:n] - g1[:n])
print("  grad perm diff: max %.3e of max|g| %.3e" % (gd.max(), np.abs(g1[:n]).max()))
