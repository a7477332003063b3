import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import ultr_oracle as O
from ultra_pytorch_amd import engine, hip_ops
def run(F, hidden, B, L, wgd):
    os.environ["ULTR_WGD"] = str(wgd)
    rng = np.random.RandomState(F + B)
    n_docs = B * L - 2
    feats = rng.uniform(-1, 1, size=(n_docs, F)).astype(np.float32)
    ids = rng.permutation(B * L); ids = np.where(ids >= n_docs, n_docs, ids).astype(np.int32).reshape(L, B)
    clicks = (rng.uniform(size=(L, B)) < 0.35).astype(np.float32); clicks[0, :] = 1.0
    params = O.init_params(F, hidden, seed=7)
    state0 = (0.01 * rng.uniform(size=params.shape)).astype(np.float32)
    ipw = np.linspace(1.0, 6.0, 12)
    shape = hip_ops.DnnShape(F, hidden, "elu")
    eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax")
    dev = lambda a, dt=torch.float32: torch.as_tensor(a).to("cuda", dt)
    p, st = dev(params), dev(state0)
    eng.train_step(p, st, dev(feats), n_docs, dev(ids, torch.int32), dev(clicks), ipw_table=dev(ipw.astype(np.float32)))
    sc = eng.read_scalars()
    torch.cuda.synchronize()
    return eng.grads.cpu().numpy().copy(), sc, eng.bwd_ws[:120].cpu().numpy().copy()
for shp in [(24, [64, 32, 32], 20, 8), (136, [256,256], 33, 10)]:
    g1, s1, w1 = run(*shp, 1); g0, s0, w0 = run(*shp, 0)
    P = len(g1) - (4 + 2*shp[3])
    print(shp, "loss", s1[0], s0[0], "norm", s1[1], s0[1], "ss", s1[7], s0[7])
    print("  tail wgd ", g1[P:P+6]); print("  tail slab", g0[P:P+6])
    d = np.abs(g1[:P]-g0[:P]); print("  max grad diff", d.max(), "at", d.argmax(), "rel", d.max()/np.abs(g0[:P]).max())
    print("  sumsq slots wgd", w1[:40].round(4))
