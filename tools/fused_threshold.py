"""GPU box: fused forward+loss+backward vs separate kernels across batch sizes (picks the eligibility threshold)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ultra_pytorch_amd import engine, hip_ops, synthetic
from ultra_pytorch_amd.ranking_model import init_flat_params
dev = torch.device("cuda")
F, hidden, L = 136, [256, 256], 10
shape = hip_ops.DnnShape(F, hidden, "elu")
ipw = torch.tensor(synthetic.load_ipw(), dtype=torch.float32, device=dev)
for B in [int(v) for v in (sys.argv[1:] or [64, 128, 256, 384, 512, 768])]:
    res = {}
    for mode in ("separate", "fused", "separate"):
        os.environ["ULTR_NO_FUSED_FB"] = "0" if mode == "fused" else "1"
        os.environ["ULTR_FB_MAX_WG_PER_CU"] = "16"  # lift the library's own threshold for the comparison
        eng = engine.StepEngine(shape, B, L, dev, algo="softmax")
        p = init_flat_params(shape, 0).to(dev); st = torch.zeros_like(p)
        f, i, y = synthetic.make_batch(np.random.RandomState(0), B, L, F)
        f, nd, i, y = torch.tensor(f, device=dev), f.shape[0], torch.tensor(i, device=dev), torch.tensor(y, device=dev)
        for k in range(50): eng.train_step(p, st, f, nd, i, y, ipw_table=ipw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(500): eng.train_step(p, st, f, nd, i, y, ipw_table=ipw)
        torch.cuda.synchronize(); res.setdefault(mode, []).append(round((time.perf_counter() - t0) / 500 * 1e6, 1))
    print("B=%4d  fused %s us   separate %s us" % (B, res["fused"], res["separate"]))
