"""Seeded sweep across the planner's rule boundaries: the library picks among three forward
kernels (dnn_fwd_kernel 16-row tiles, dnn_fwdw_kernel wide tiles, the per-layer launches of ultr_dnn_big.hip), four backward kernels
(dnn_bwd_kernel, dnn_bwd2_kernel, dnn_bwdw_kernel, per-layer), two weight-gradient kernels and the fused forward + loss + backward
launch by SHAPE RULES (ultr_make_dnn_plan, fwd_wide_plan, bwd_wide_plan, "measured" thresholds).  ~70 shapes drawn by
RandomState(0) around every boundary - rows per workgroup 16 / 17, 48 / 49, 64 / 65 (N = 256 CUs x R +- a few rows), layer widths
255 / 256 / 257 and 511 / 512 / 513, feature sizes that are not multiples of 4, zero to four hidden layers, list sizes 1, 9, 10, 17,
100, 101, PAD documents (rank_list_size < max_candidate_num), training and evaluation - each runs ONE product step
(engine.StepEngine.train_step = ultr_train_step) against oracle.ultr_oracle at the golden tolerances (loss 1e-5, scores 1e-5,
gradients 1e-5 (|g| + sum |terms|) per entry), or one evaluation forward (scores 1e-5).  The sweep asserts that it reached every kernel
family (the planners ultr_dnn_forward_tile_rows / ultr_dnn_backward_tile_rows say what a shape takes).
Reference: DNN.py:18-56 (any widths, any depth), base_algorithm.py:118-154, ipw_rank.py:102-182."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

WIDTHS = [255, 256, 257, 511, 512, 513, 32, 64, 96, 128, 40]
FEATS = [13, 24, 46, 136, 137, 220, 700]
LISTS = [1, 9, 10, 17, 100, 101]
ROWS_PER_WG = [16, 17, 48, 49, 64, 65, 4, 30]
ACTS = ["elu", "relu", "tanh", "sigmoid"]


def draw_cases():
    rng = np.random.RandomState(0)
    cases = []
    for k in range(72):
        L = int(LISTS[k % len(LISTS)])
        R = int(ROWS_PER_WG[(k // len(LISTS)) % len(ROWS_PER_WG)])
        # N = 256 compute units x R rows, a few rows either side (the planners round per 256 workgroups)
        target = 256 * R + int(rng.randint(-40, 41))
        B = max(1, int(round(target / L)))
        nh = int(rng.randint(0, 5))
        hidden = [int(WIDTHS[rng.randint(len(WIDTHS))]) for _ in range(nh)]
        F = int(FEATS[rng.randint(len(FEATS))])
        # keep one case's oracle step under ~0.5 s of CPU: cap rows x sum(in x out)
        dims = [F] + hidden + [1]
        work = B * L * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
        while work > 1.2e9 and hidden:
            hidden = [max(32, h // 2) if h not in (255, 257, 511, 513) else h // 2 + 1 for h in hidden]
            dims = [F] + hidden + [1]
            work = B * L * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
            if work > 1.2e9 and F > 136:
                F = 136
        n_pad = int(rng.randint(0, min(4, L))) if L > 1 else 0
        algo = "dla" if k % 5 == 4 else "softmax"
        train = k % 7 != 6
        cases.append(dict(k=k, F=F, hidden=hidden, B=B, L=L, act=ACTS[rng.randint(4)], n_pad=n_pad, algo=algo, train=train))
    return cases


CASES = draw_cases()
SEEN = {}


def families(c):
    from ultra_pytorch_amd import _lib, hip_ops
    lib = _lib.load()
    shape = hip_ops.DnnShape(c["F"], c["hidden"], c["act"])
    n = c["B"] * c["L"]
    fwd = lib.ultr_dnn_forward_tile_rows(shape.desc, n, 1 if c["train"] else 0)
    name = lambda code: "wide" if code >= 1000 else ("per_layer" if code == 0 else "tile%d" % code)
    if not c["train"]:
        return ["eval_fwd_" + name(fwd)]
    bwd = lib.ultr_dnn_backward_tile_rows(shape.desc, n)
    return ["fwd_" + name(fwd), "bwd_" + name(bwd)]


@pytest.mark.parametrize("c", CASES, ids=lambda c: "%02d_F%d_%s_B%dxL%d_%s_%s" % (c["k"], c["F"], "x".join(map(str, c["hidden"])) or "linear", c["B"], c["L"],
                                                                                  c["algo"], "train" if c["train"] else "eval"))
def test_one_step_against_the_oracle(c):
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops, synthetic
    from tests.hipref import dev
    F, hidden, B, L, act = c["F"], c["hidden"], c["B"], c["L"], c["act"]
    rng = np.random.RandomState(1000 + c["k"])
    feats, ids, y = synthetic.make_batch(rng, B, L, F, clicks=True, n_pad=c["n_pad"])
    params = O.init_params(F, hidden, seed=7 + c["k"])
    for n, s, o in O.param_layout(F, hidden):  # LayerNorm affine parameters away from (1, 0)
        if "layer_norm" in n:
            params[o:o + int(np.prod(s))] += rng.normal(scale=0.2, size=int(np.prod(s))).astype(np.float32)
    for fam in families(c):
        SEEN[fam] = SEEN.get(fam, 0) + 1
    shape = hip_ops.DnnShape(F, hidden, act)
    if not c["train"]:
        ev = engine.EvalEngine(shape, B, L, torch.device("cuda"))
        scores, _ = ev.run(dev(params), dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y))
        ref = O.ranking_scores(torch.from_numpy(params), F, hidden, feats, ids, act).numpy()
        np.testing.assert_allclose(scores.cpu().numpy(), ref, atol=1e-5, rtol=1e-5)
        return
    ipw = np.linspace(1.0, 6.0, 12).astype(np.float32)
    x = O.gather_rows(feats, ids).numpy()
    if c["algo"] == "softmax":
        ref = O.train_step_softmax(params, np.zeros_like(params), F, hidden, feats, ids.astype(np.int64), y, ipw_list=ipw, act=act)
        eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax")
        p, st = dev(params.copy()), dev(np.zeros_like(params))
        sc = eng.train_step(p, st, dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y), ipw_table=dev(ipw))
        torch.cuda.synchronize()
        scal = sc.cpu().numpy()
        assert abs(scal[0] - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"])), (scal[0], ref["loss"])
        np.testing.assert_allclose(eng.scores.cpu().numpy(), ref["scores"], atol=1e-5, rtol=1e-5)
        g = eng.grads[:shape.n_params].cpu().numpy() / float(scal[3])
        gref = ref["grads"]
        # per-entry bar: 1e-5 (|g| + sum |terms|) - the terms through the same d(loss)/d(score) (tests/test_gpu_full_size.py)
        lab = torch.from_numpy(np.ascontiguousarray(y.T)).float()
        s_t = torch.from_numpy(ref["scores"]).clone().requires_grad_(True)
        (dsc,) = torch.autograd.grad(O.softmax_loss(s_t, lab, torch.from_numpy(ref["pw"]) if ref["pw"] is not None else None), s_t)
        terms = O.dnn_backward_manual(params, F, hidden, x, dsc.numpy().reshape(-1), act, abs_terms=True)
        d = np.abs(g - gref)
        if act == "relu":
            # ReLU's derivative is a step: among ~10^6 pre-activations a few sit within an ulp of zero and land on the other side of
            # it under another summation order (the reference's own 1-thread and 8-thread runs do the same), each flip moves a few
            # gradient entries by a whole term (layers below a flipped unit: every entry a little).  Held to 2e-3 of the largest entry.
            assert d.max() <= 2e-3 * np.abs(gref).max(), (float(d.max()), float(np.abs(gref).max()))
        else:
            assert (d <= 1e-5 * (np.abs(gref) + terms) + 1e-12).all(), float((d / np.maximum(np.abs(gref) + terms, 1e-30)).max())
        eng.close()
    else:
        prop = (0.1 * rng.randn(L + 1)).astype(np.float32)
        ref = O.dla_step(params, prop, F, hidden, feats, ids.astype(np.int64), y, act=act)
        eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="dla")
        p, aux = dev(params.copy()), dev(prop.copy())
        sc = eng.train_step(p, None, dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y), aux=aux)
        torch.cuda.synchronize()
        scal = sc.cpu().numpy()
        assert abs(scal[0] - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"])), (scal[0], ref["loss"])
        np.testing.assert_allclose(eng.scores.cpu().numpy(), ref["scores"], atol=1e-5, rtol=1e-5)
        # DLA's update is sign-like (dla.py:141-177): compare the propensity parameters (smooth) and the ranker's parameters where
        # the gradient is not ~0
        np.testing.assert_allclose(aux.cpu().numpy(), ref["prop_params"], atol=2e-5)
        if "grads" in ref:
            gref = ref["grads"]
            sel = np.abs(gref) > 1e-3 * np.abs(gref).max()
            np.testing.assert_allclose(p.cpu().numpy()[sel], ref["params"][sel], rtol=1e-4, atol=1e-5)
        eng.close()


def test_the_sweep_reached_every_kernel_family():
    """(runs after the cases above: pytest keeps file order)"""
    need = ["fwd_tile16", "fwd_wide", "bwd_wide", "eval_fwd_tile16", "eval_fwd_wide"]
    missing = [f for f in need if SEEN.get(f, 0) == 0]
    assert not missing, (missing, SEEN)
    assert any(k.startswith("bwd_tile") for k in SEEN), SEEN
    print("kernel families reached:", dict(sorted(SEEN.items())))
