import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import ultr_oracle as O
from ultra_pytorch_amd import engine, hip_ops, synthetic, _lib
from tests.hipref import dev
import importlib.util
spec = importlib.util.spec_from_file_location("sw", "tests/test_gpu_planner_sweep.py"); sw = importlib.util.module_from_spec(spec); spec.loader.exec_module(sw)
which = [int(a) for a in sys.argv[1:] if a.isdigit()] or [5]
custom = [a for a in sys.argv[1:] if not a.isdigit()]  # F:h1,h2,..:B:L:act
cases = [c for c in sw.CASES if c["k"] in which] if not custom else []
for a in custom:
    F_, hs, B_, L_, act_ = a.split(":")
    cases.append(dict(k=5, F=int(F_), hidden=[int(h) for h in hs.split(",") if h], B=int(B_), L=int(L_), act=act_, n_pad=0, algo="softmax", train=True))
for c in cases:
    F, hidden, B, L, act = c["F"], c["hidden"], c["B"], c["L"], c["act"]
    rng = np.random.RandomState(1000 + c["k"])
    feats, ids, y = synthetic.make_batch(rng, B, L, F, clicks=True, n_pad=c["n_pad"])
    params = O.init_params(F, hidden, seed=7 + c["k"])
    for n, s, o in O.param_layout(F, hidden):
        if "layer_norm" in n:
            params[o:o + int(np.prod(s))] += rng.normal(scale=0.2, size=int(np.prod(s))).astype(np.float32)
    print(c, sw.families(c))
    shape = hip_ops.DnnShape(F, hidden, act)
    ipw = np.linspace(1.0, 6.0, 12).astype(np.float32)
    x = O.gather_rows(feats, ids).numpy()
    ref = O.train_step_softmax(params, np.zeros_like(params), F, hidden, feats, ids.astype(np.int64), y, ipw_list=ipw, act=act)
    for env in ({},):
        for k in ("ULTR_FB_H3", "ULTR_FWD_H3", "ULTR_BWD_H3", "ULTR_WG_H3"): os.environ.pop(k, None)
        os.environ.update(env)
        eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax")
        p, st = dev(params.copy()), dev(np.zeros_like(params))
        sc = eng.train_step(p, st, dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y), ipw_table=dev(ipw))
        torch.cuda.synchronize()
        scal = sc.cpu().numpy()
        g = eng.grads[:shape.n_params].cpu().numpy() / float(scal[3])
        gref = ref["grads"]
        lab = torch.from_numpy(np.ascontiguousarray(y.T)).float()
        s_t = torch.from_numpy(ref["scores"]).clone().requires_grad_(True)
        (dsc,) = torch.autograd.grad(O.softmax_loss(s_t, lab, torch.from_numpy(ref["pw"])), s_t)
        terms = O.dnn_backward_manual(params, F, hidden, x, dsc.numpy().reshape(-1), act, abs_terms=True)
        d = np.abs(g - gref)
        print(" env", env, "loss", scal[0], ref["loss"], "score err", np.abs(eng.scores.cpu().numpy() - ref["scores"]).max())
        worst = []
        for n, s, o in O.param_layout(F, hidden):
            k = int(np.prod(s))
            r = (d[o:o + k] / np.maximum(np.abs(gref[o:o + k]) + terms[o:o + k], 1e-30))
            worst.append((float(r.max()), n))
        print("   worst tensors:", ["%s %.1e" % (n, v) for v, n in sorted(worst, reverse=True)[:4]])
