"""GPU box: SetRank's training step (BASELINE config 5 at full size, and a ragged smaller shape) run many times from the same state; counts
the runs whose scores / gradients / updated parameters differ bitwise from the first - the persistent fused kernels, the attention kernels and
the one-launch fold keep fixed-order sums, so a deterministic library prints 0 everywhere.  (tools/h3_stress.py is the DNN counterpart.)
    python tools/sr_stress.py [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from ultra_pytorch_amd import engine, hip_ops, synthetic  # noqa: E402
from ultra_pytorch_amd.ranking_model.SetRank import init_setrank_params  # noqa: E402


def stress(B, L, F, dm, H, nl, dff, reps, att="fp32"):
    shape = hip_ops.SetRankShape(F, dm, H, nl, dff, attention_dtype=att)
    rng = np.random.RandomState(5)
    feats, ids, y = synthetic.make_batch(rng, B, L, F, n_pad=3)
    p0 = init_setrank_params(shape, seed=3).numpy()
    p0 += rng.normal(scale=0.02, size=p0.shape).astype(np.float32)
    dev = lambda a, dt=torch.float32: torch.as_tensor(a).to("cuda", dt)
    eng = engine.SetRankStepEngine(shape, B, L, torch.device("cuda"), algo="softmax", learning_rate=0.05, max_gradient_norm=5.0)
    f, i, yy = dev(feats), dev(ids, torch.int32), dev(y)
    first, bad = None, [0, 0, 0]
    for r in range(reps):
        params, state = dev(p0.copy()), dev(np.zeros_like(p0))
        eng.train_step(params, state, f, feats.shape[0], i, yy)
        torch.cuda.synchronize()
        cur = (eng.scores.clone(), eng.grads[:shape.n_params].clone(), params.clone())
        assert bool(torch.isfinite(cur[1]).all())
        if first is None:
            first = cur
            continue
        for k in range(3):
            bad[k] += 0 if torch.equal(cur[k], first[k]) else 1
    print("B%d L%d F%d d%d H%d layers %d dff %d attention %s: runs that differ from the first - scores %d, gradients %d, parameters %d of %d" %
          (B, L, F, dm, H, nl, dff, att, bad[0], bad[1], bad[2], reps - 1), flush=True)
    return sum(bad)


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    tot = stress(1024, 100, 220, 256, 8, 2, 64, reps)
    tot += stress(1024, 100, 220, 256, 8, 2, 64, max(reps // 4, 5), att="fp16")
    tot += stress(37, 30, 136, 256, 8, 2, 64, reps)
    tot += stress(200, 50, 220, 256, 8, 2, 64, reps)
    tot += stress(64, 40, 24, 128, 4, 1, 128, reps)
    print("TOTAL", tot)
    sys.exit(1 if tot else 0)
