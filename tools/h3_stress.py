"""GPU box: run the SAME training step many times from the same state and count launches whose gradients / scores differ
bitwise from the first one - per kernel family (fused small-batch kernel, separate forward / backward kernels), with the
split-half (fp16 hi/lo) products on and off.  A deterministic library gives 0 everywhere.
    python tools/h3_stress.py [reps]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from ultra_pytorch_amd import engine, hip_ops, synthetic
from ultra_pytorch_amd.ranking_model import init_flat_params

def stress(F, hidden, B, L, env, reps, algo="softmax"):
    for k in ("ULTR_NO_FUSED_FB", "ULTR_FB_H3", "ULTR_FWD_H3", "ULTR_BWD_H3"):
        os.environ.pop(k, None)
    os.environ.update(env)
    shape = hip_ops.DnnShape(F, hidden, "elu")
    rng = np.random.RandomState(5)
    feats, ids, y = synthetic.make_batch(rng, B, L, F)
    ipw = np.asarray(synthetic.load_ipw(), np.float32)
    p0 = init_flat_params(shape, seed=3).numpy()
    dev = lambda a, dt=torch.float32: torch.as_tensor(a).to("cuda", dt)
    eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo=algo, learning_rate=0.05, max_gradient_norm=5.0)
    f, i, yy, tab = dev(feats), dev(ids, torch.int32), dev(y), dev(ipw)
    first, bad_g, bad_s, worst = None, 0, 0, 0.0
    for r in range(reps):
        params, state = dev(p0.copy()), dev(np.zeros_like(p0))
        eng.train_step(params, state, f, feats.shape[0], i, yy, ipw_table=tab)
        torch.cuda.synchronize()
        g, s = eng.grads[:shape.n_params].clone(), eng.scores.clone()
        if first is None:
            first = (g, s)
            continue
        if not torch.equal(g, first[0]):
            bad_g += 1
            worst = max(worst, float((g - first[0]).abs().max() / first[0].abs().max()))
        if not torch.equal(s, first[1]):
            bad_s += 1
    print("%-34s %-52s launches with different grads %3d / %d (worst rel %.1e), scores %3d" %
          ("F%d %s B%d L%d" % (F, hidden, B, L), " ".join("%s=%s" % kv for kv in sorted(env.items())), bad_g, reps - 1, worst, bad_s), flush=True)
    return bad_g + bad_s

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    tot = 0
    if os.environ.get("H3_STRESS_ONLY_BWD") == "1":
        tot += stress(136, [256, 256], 256, 10, {"ULTR_NO_FUSED_FB": "1"}, reps)
        tot += stress(136, [512, 256, 128], 512, 20, {}, reps)
        print("TOTAL", tot)
        sys.exit(0)
    for env in ({}, {"ULTR_FB_H3": "0"}):
        tot += stress(136, [256, 256], 256, 10, env, reps)
    for env in ({"ULTR_NO_FUSED_FB": "1"}, {"ULTR_NO_FUSED_FB": "1", "ULTR_BWD_H3": "0"}, {"ULTR_NO_FUSED_FB": "1", "ULTR_BWD_H3": "0", "ULTR_FWD_H3": "0"}):
        tot += stress(136, [256, 256], 256, 10, env, reps)
    for env in ({}, {"ULTR_BWD_H3": "0"}, {"ULTR_BWD_H3": "0", "ULTR_FWD_H3": "0"}):
        tot += stress(136, [512, 256, 128], 512, 20, env, reps)
    print("TOTAL", tot)
