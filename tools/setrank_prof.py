"""GPU box: a few SetRank steps at config 5 (run under rocprofv3 --kernel-trace --stats to see where the time goes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ultra_pytorch_amd import engine, hip_ops, synthetic
from ultra_pytorch_amd.ranking_model.SetRank import init_setrank_params
B, L, F = int(os.environ.get("SR_B", "1024")), 100, 220
dev = torch.device("cuda")
shape = hip_ops.SetRankShape(F, 256, 8, 2, 64)
eng = engine.SetRankStepEngine(shape, B, L, dev, algo="softmax", learning_rate=0.05)
p = init_setrank_params(shape, 0).to(dev)
st = torch.zeros_like(p)
f, i, y = synthetic.make_batch(np.random.RandomState(0), B, L, F)
f, nd, i, y = torch.tensor(f, device=dev), f.shape[0], torch.tensor(i, device=dev), torch.tensor(y, device=dev)
ipw = torch.tensor(synthetic.load_ipw(), dtype=torch.float32, device=dev)
for k in range(6):
    eng.train_step(p, st, f, nd, i, y, ipw_table=ipw)
torch.cuda.synchronize()
