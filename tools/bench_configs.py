"""GPU box: time one training step of every BASELINE config (not bench lines)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ultra_pytorch_amd import engine, hip_ops, synthetic
from ultra_pytorch_amd.ranking_model import init_flat_params
CONFIGS = {"cfg2 IPW F136 L10 B256 [256,256]": (136, [256, 256], 256, 10, "softmax", 0.05),
           "cfg3 DLA F136 L20 B512 [512,256,128]": (136, [512, 256, 128], 512, 20, "dla", 0.05),
           "cfg4 PairDebias F700 L50 B256 [512,256,128]": (700, [512, 256, 128], 256, 50, "pairdebias", 0.005),
           "cfg4 LambdaRank F700 L50 B256 [512,256,128]": (700, [512, 256, 128], 256, 50, "lambdarank", 0.05),
           "cfg2x4 IPW F136 L10 B1024 [256,256]": (136, [256, 256], 1024, 10, "softmax", 0.05)}
if os.environ.get("ULTR_BENCH_BIG", "0") == "1":  # throughput end of the batch axis (saved activations: 0.5 GB at B = 65536)
    CONFIGS = {"cfg2x32 IPW F136 L10 B8192 [256,256]": (136, [256, 256], 8192, 10, "softmax", 0.05),
               "cfg2x256 IPW F136 L10 B65536 [256,256]": (136, [256, 256], 65536, 10, "softmax", 0.05),
               "cfg3x16 DLA F136 L20 B8192 [512,256,128]": (136, [512, 256, 128], 8192, 20, "dla", 0.05)}
dev = torch.device("cuda")
for name, (F, hidden, B, L, algo, lr) in CONFIGS.items():
    shape = hip_ops.DnnShape(F, hidden, "elu")
    eng = engine.StepEngine(shape, B, L, dev, algo=algo, learning_rate=lr)
    p = init_flat_params(shape, 0).to(dev)
    st = None if algo == "dla" else torch.zeros_like(p)
    rng = np.random.RandomState(0)
    pool = []
    for _ in range(4):
        f, i, y = synthetic.make_batch(rng, B, L, F, clicks=(algo != "lambdarank"))
        pool.append((torch.tensor(f, device=dev), f.shape[0], torch.tensor(i, device=dev), torch.tensor(y, device=dev)))
    aux = None
    if algo == "dla":
        aux = torch.zeros(L + 1, device=dev)
    elif algo != "softmax":
        aux = torch.ones(2 * L, device=dev)
    ipw = torch.tensor(synthetic.load_ipw(), dtype=torch.float32, device=dev) if algo == "softmax" else None
    def step(k):
        f, nd, i, y = pool[k % 4]
        return eng.train_step(p, st, f, nd, i, y, aux=aux, ipw_table=ipw)
    for k in range(30): step(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 300
    for k in range(n): step(k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    dims = [(F, hidden[0])] + list(zip(hidden[:-1], hidden[1:])) + [(hidden[-1], 1)]
    S = sum(a * b for a, b in dims)
    flops = B * L * (6 * S - 2 * F * hidden[0])
    import ctypes
    from ultra_pytorch_amd import _lib
    lib = _lib.load()
    tot, cnt = (ctypes.c_double * 8)(), (ctypes.c_int64 * 8)()
    lib.ultr_prof_set_stride(1)
    lib.ultr_prof_enable(0xBF, 8 * 40)
    for k in range(40): step(k)
    torch.cuda.synchronize()
    lib.ultr_prof_collect(tot, cnt)
    lib.ultr_prof_enable(0, 0)
    kn = ["fwd", "loss", "bwd", "wgrad", "reduce", "update", "ndcg", "fused"]
    kus = {kn[k]: round(1e3 * tot[k] / cnt[k], 1) for k in range(8) if cnt[k] > 0}
    print("%-46s %8.1f us/step %10.0f q/s  %6.2f TFLOP/s (%.1f%% of fp32 MFMA peak)  loss %.4f" % (name, dt * 1e6, B / dt, flops / dt / 1e12, 100 * flops / dt / 157.3e12, float(eng.scalars[0])), kus)


# ---- config 5: SetRank, F220 L100 B1024, d_model 256, 8 heads x 32, 2 layers, dff 64, IPW ---------------------
from ultra_pytorch_amd.ranking_model.SetRank import init_setrank_params
for name, (F, B, L) in {"cfg5 IPW+SetRank F220 L100 B1024 d256 h8 l2 dff64": (220, 1024, 100),
                        "cfg5/4 IPW+SetRank F220 L100 B256": (220, 256, 100)}.items():
    shape = hip_ops.SetRankShape(F, 256, 8, 2, 64)
    eng = engine.SetRankStepEngine(shape, B, L, dev, algo="softmax", learning_rate=0.05)
    p = init_setrank_params(shape, 0).to(dev)
    st = torch.zeros_like(p)
    rng = np.random.RandomState(0)
    f, i, y = synthetic.make_batch(rng, B, L, F)
    f, nd, i, y = torch.tensor(f, device=dev), f.shape[0], torch.tensor(i, device=dev), torch.tensor(y, device=dev)
    ipw = torch.tensor(synthetic.load_ipw(), dtype=torch.float32, device=dev)
    for k in range(3): eng.train_step(p, st, f, nd, i, y, ipw_table=ipw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for k in range(n): eng.train_step(p, st, f, nd, i, y, ipw_table=ipw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    # per token MACs: input FFN + 2 x (attention 2*L*d + dense d*d + FFN 2*d*dff) + output FFN; x3 for fwd+bwd
    mac = F * 64 + 64 * 256 + 2 * (2 * L * 256 + 256 * 256 + 2 * 256 * 64) + 256 * 64 + 64
    flops = 3 * 2.0 * mac * B * L
    print("%-52s %9.1f us/step %9.0f q/s  %6.2f TFLOP/s  loss %.4f" % (name, dt * 1e6, B / dt, flops / dt / 1e12, float(eng.scalars[0])))
