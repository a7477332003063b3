"""GPU parity tests proper: the HIP path (through the C ABI) against
  (1) the golden vectors captured from the reference, stage by stage, and
  (2) the oracle on seeded inputs at other shapes (incl. BASELINE configs' layer shapes).
Tolerances (fp32): scores 1e-5 abs; loss 1e-5 (relative for |loss| > 1); gradients 1e-5 rel + 1e-6*max|g| abs;
EM state 1e-6; NDCG values 1e-6 with the identical permutation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import margins  # noqa: E402
from tests.hipref import HipRun, dev, load_golden  # noqa: E402

ALGO = {"na": "softmax", "ipw": "softmax", "dla": "dla", "pairdebias": "pairdebias", "lambdarank": "lambdarank",
        "regem": "regem"}
TRAIN_CASES = ["na_tiny", "ipw_tiny", "dla_tiny", "dla_sigmoid", "dla_sigmoid_odd", "pairdebias_tiny", "lambdarank_tiny", "ipw_odd", "dla_odd",
               "pairdebias_odd", "lambdarank_odd", "na_linear", "ipw_relu", "ipw_sgd", "ipw_cfg2", "regem_tiny", "regem_odd",
               # the remaining activations (base_ranking_model.py:63-69) and l2_loss > 0 (clip skipped except for DLA, Appendix A.8)
               "na_tanh", "na_sigmoid", "ipw_l2", "na_l2", "dla_l2", "pairdebias_l2", "regem_l2"]


def act_of(name):
    return {"ipw_relu": "relu", "na_tanh": "tanh", "na_sigmoid": "sigmoid"}.get(name, "elu")  # activation_func of the fixture's model


def gtol(g, name=""):
    """1e-5 relative + 1e-6*max|g| absolute.  The *_odd fixtures have LayerNorms over 3-6 units (rstd up to 54):
    fp32 evaluation-order noise in their inputs is amplified ~10x, for torch-CPU and for the HIP path alike
    (tools/diag_precision.py: every stage is individually as accurate as torch fp32 against an fp64 evaluation),
    so those ill-conditioned toy nets get 2e-5 / 1e-5*max|g| (round 5: closed from 1e-4 - measured <= 6.2e-6 x max|g|,
    profiles/r04_parity_margins.json)."""
    if name.endswith("_odd"):
        return dict(rtol=2e-5, atol=1e-5 * max(1.0, float(np.abs(g).max())))
    return dict(rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(g).max())))


def make_run(m, name):
    kw = dict(learning_rate=m["lr"], max_gradient_norm=m["max_gradient_norm"])
    if "sgd" in name:
        kw["optimizer"] = "sgd"
    hp = dict(kv.split("=") for kv in m.get("algo_hparams", "").split(",") if kv)
    kw["l2_loss"] = float(hp.get("l2_loss", 0.0))
    if m["algo"] == "dla":  # dla.py:70-77 hparams
        kw["logits_to_prob"] = hp.get("logits_to_prob", "softmax")
        kw["ranker_loss_weight"] = float(hp.get("ranker_loss_weight", 1.0))
        kw["propensity_learning_rate"] = float(hp.get("propensity_learning_rate", -1.0))
    return HipRun(m["F"], m["hidden"] or [], m["B"], m["L"], algo=ALGO[m["algo"]], act=act_of(name), **kw)


def gscale_and_loss(algo, tail, rw=1.0):
    loss_sum, D, loss2, D2 = [float(x) for x in tail[:4]]
    if algo in ("na", "ipw"):
        return 1.0 / D, loss_sum / D
    if algo == "dla":
        return rw / D, loss2 / D2 + rw * loss_sum / D
    if algo == "pairdebias":
        return 1.0, loss_sum
    return 1.0 / D, loss_sum / D


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_golden_train_step(name):
    _golden_train_step(name)


def test_golden_train_step_on_the_fp32_mfma_plan(mfma_mode):
    """The reference's config-2-shaped fixture (the only golden case whose layers are wide enough for the split-half products)
    under BOTH arithmetic plans; test_golden_train_step above runs the default one for every fixture."""
    _golden_train_step("ipw_cfg2")


def _golden_train_step(name):
    d, m = load_golden(name)
    run = make_run(m, name)
    L = m["L"]
    l2 = float(run.eng.udesc.l2_loss)
    for t in range(m["n_steps"]):
        p = "s%d_" % t
        run.set_inputs(d[p + "features"], d[p + "docids"], d[p + "labels"])
        # --- forward
        scores = run.forward(d[p + "pre_params"])
        np.testing.assert_allclose(scores, d[p + "scores"], atol=1e-5, rtol=0, err_msg="scores")
        # --- loss (fed with the HIP scores)
        aux = None
        if m["algo"] == "dla":
            aux = d[p + "pre_prop_params"]
        elif m["algo"] in ("pairdebias", "lambdarank"):
            aux = np.concatenate([d[p + "pre_t_plus"].ravel(), d[p + "pre_t_minus"].ravel()])
        elif m["algo"] == "regem":
            aux = d[p + "pre_propensity"].ravel()
        ipw = d["ipw_list"] if m["algo"] == "ipw" else None
        # RegressionEM: teacher-forced with the uniforms the reference drew
        ds, tail = run.loss(aux=aux, ipw_table=ipw, uniforms=d[p + "uniforms"] if m["algo"] == "regem" else None)
        if m["algo"] == "regem":
            # the pseudo-labels are the only discrete quantity: dscores x D = sigmoid(s) - y  =>  y = sigmoid(s) - ds
            y = 1.0 / (1.0 + np.exp(-scores.astype(np.float64))) - ds
            np.testing.assert_allclose(y, d[p + "ranker_labels"], atol=1e-5)
        gs, loss = gscale_and_loss(m["algo"], tail, rw=run.eng.udesc.ranker_loss_weight)
        ref_loss = float(d[p + "loss"])
        # l2_loss > 0: loss += l2 * sum p^2 / 2 and g += l2 * p (inside rank_loss, i.e. x ranker_loss_weight, for DLA) - both are
        # formed by the update launch; here they are added to the stage outputs for the comparison
        lam = l2 * (run.eng.udesc.ranker_loss_weight if m["algo"] == "dla" else 1.0)
        p0 = d[p + "pre_params"].astype(np.float64)
        loss += lam * 0.5 * float((p0 * p0).sum())
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), ("loss", loss, ref_loss)
        # --- backward
        g, tail2 = run.backward()
        np.testing.assert_allclose(tail2, tail, rtol=1e-6, atol=1e-6)
        gref = d[p + "grads"]
        if name.endswith("_odd"):  # widened band (gtol): keep the measured figure visible
            gdiff = np.abs(g * gs + lam * d[p + "pre_params"] - gref)
            margins.check("golden/" + name, "s%d_grads_max_abs_diff_over_max_abs_g" % t, gdiff.max() / max(1.0, float(np.abs(gref).max())))
            margins.check("golden/" + name, "s%d_scores_max_abs_diff" % t, np.abs(scores - d[p + "scores"]).max())
        np.testing.assert_allclose(g * gs + lam * d[p + "pre_params"], gref, err_msg="grads", **gtol(gref, name))
        # --- update
        state = d[p + "pre_adagrad"] if (p + "pre_adagrad") in d.files else None
        params, state2, aux2, sc = run.update(state)
        assert abs(sc[0] - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
        if l2 > 0 and m["algo"] != "dla":
            # the reference's clip_grad_norm_ got an exhausted generator: it measured nothing and clipped nothing, although
            # the gradient norm exceeds max_gradient_norm in these fixtures
            assert float(d[p + "norm"]) == 0.0 and sc[2] == 1.0 and sc[1] > m["max_gradient_norm"]
            assert abs(sc[1] - float(np.linalg.norm(gref.astype(np.float64)))) <= 1e-5 * sc[1]
        else:
            assert abs(sc[1] - float(d[p + "norm"])) <= 1e-5 * max(1.0, float(d[p + "norm"]))
        sel = np.abs(gref) > 1e-6 * max(1.0, float(np.abs(gref).max()))
        np.testing.assert_allclose(params[sel], d[p + "post_params"][sel], atol=5e-6, rtol=1e-5, err_msg="params")
        if m["algo"] in ("pairdebias", "lambdarank"):
            np.testing.assert_allclose(aux2[:L], d[p + "post_t_plus"].ravel(), atol=1e-6)
            np.testing.assert_allclose(aux2[L:], d[p + "post_t_minus"].ravel(), atol=1e-6)
        if m["algo"] == "regem":
            np.testing.assert_allclose(aux2, d[p + "post_propensity"].ravel(), atol=1e-6)
        if m["algo"] == "dla":
            np.testing.assert_allclose(aux2, d[p + "post_prop_params"], atol=1e-6)
            assert abs(sc[6] - float(d[p + "prop_norm"])) < 1e-6
            assert abs(sc[4] - float(d[p + "rank_loss"])) < 1e-5 and abs(sc[5] - float(d[p + "exam_loss"])) < 1e-5
        if state is not None and "sgd" not in name:
            ref_state = d[p + "post_adagrad"]
            # s' = s + g^2 with g at the 1e-5 bar: 2e-5 relative (+ 2e-6 * max for elements with g ~ 0); the ill-conditioned
            # *_odd nets carry their 2e-5 gradient bar (gtol) into the accumulator
            np.testing.assert_allclose(state2, ref_state, rtol=4e-5 if name.endswith("_odd") else 2e-5,
                                       atol=2e-6 * float(ref_state.max()))


@pytest.mark.parametrize("name", ["valid_tiny", "valid_odd"])
def test_golden_validation(name):
    from ultra_pytorch_amd import engine, hip_ops
    d, m = load_golden(name)
    shape = hip_ops.DnnShape(m["F"], m["hidden"], "elu")
    params = dev(d["params"])
    for b in range(int(d["n_batches"])):
        p = "b%d_" % b
        docids = d[p + "docids"]
        L, B = docids.shape
        ev = engine.EvalEngine(shape, B, L, torch.device("cuda"), topn=m["topn"])
        feats = d[p + "features"]
        scores, nd = ev.run(params, dev(feats), feats.shape[0], dev(docids, torch.int32), dev(d[p + "labels"]))
        torch.cuda.synchronize()
        np.testing.assert_allclose(scores.cpu().numpy(), d[p + "scores"], atol=1e-5)   # UNMASKED scores returned
        np.testing.assert_allclose(ev.masked.cpu().numpy(), d[p + "masked_scores"], atol=1e-5)
        # bit-exact ordering: identical permutation wherever the sort key is unique; among exactly tied keys (the
        # -100000 padding scores) torch's unstable sort order is arbitrary, so there only the key sequence must match
        order, ref_order = ev.order.cpu().numpy().astype(np.int64), d[p + "argsort_desc"].astype(np.int64)
        key = d[p + "masked_scores"]
        k_ours, k_ref = np.take_along_axis(key, order, 1), np.take_along_axis(key, ref_order, 1)
        np.testing.assert_array_equal(k_ours, k_ref)
        tied = np.zeros_like(k_ref, dtype=bool)
        tied[:, 1:] |= k_ref[:, 1:] == k_ref[:, :-1]
        tied[:, :-1] |= k_ref[:, :-1] == k_ref[:, 1:]
        np.testing.assert_array_equal(order[~tied], ref_order[~tied])
        assert sorted(order[0].tolist()) == list(range(L))
        for i, n in enumerate(m["topn"]):
            assert abs(float(nd[i]) - float(d[p + "metric_ndcg_%d" % n])) < 1e-6


# ------------------------------------------------------------------------------------------------
# oracle parity at other shapes (all tile paths: R=16/32, CT=1/2/4, vec/scalar, ragged edges)
# ------------------------------------------------------------------------------------------------
SHAPES = [
    # F, hidden, B, L
    (136, [256, 256], 64, 10),        # config 2 layers, smaller batch
    (136, [512, 256, 128], 48, 20),   # config 3 layers
    (700, [512, 256, 128], 16, 50),   # config 4 layers
    (136, [256, 256], 900, 10),       # > 512 row blocks -> 32-row workgroups
    (20, [64], 7, 3),
    (33, [17, 9], 5, 4),              # nothing aligned
    (136, [], 32, 10),                # Linear model
]


def synth(F, B, L, seed):
    rng = np.random.RandomState(seed)
    n_docs = B * L - 3 if B * L > 8 else B * L
    feats = rng.uniform(-1, 1, size=(n_docs, F)).astype(np.float32)
    ids = rng.permutation(B * L)
    ids = np.where(ids >= n_docs, n_docs, ids).astype(np.int32).reshape(L, B)  # a few PAD docs
    labels = (rng.uniform(size=(L, B)) < 0.3).astype(np.float32)
    labels[0, :] = 1.0
    return feats, ids, labels


@pytest.mark.parametrize("F,hidden,B,L", SHAPES)
def test_oracle_forward_backward(F, hidden, B, L):
    from oracle import ultr_oracle as O
    feats, ids, labels = synth(F, B, L, 5)
    params = O.init_params(F, hidden, seed=3)
    rng = np.random.RandomState(9)
    # non-trivial LayerNorm affine parameters
    for name, shape, off in O.param_layout(F, hidden):
        if "layer_norm" in name:
            n = int(np.prod(shape))
            params[off:off + n] += rng.uniform(-0.3, 0.3, size=n).astype(np.float32)
    run = HipRun(F, hidden, B, L)
    run.set_inputs(feats, ids, labels)
    scores = run.forward(params)
    p = torch.tensor(params, requires_grad=True)
    ref = O.ranking_scores(p, F, hidden, feats, ids)
    np.testing.assert_allclose(scores, ref.detach().numpy(), atol=1e-5, rtol=1e-5)
    ds = rng.normal(size=(B, L)).astype(np.float32)
    (gref,) = torch.autograd.grad((ref * torch.tensor(ds)).sum(), p)
    g, _ = run.backward(dscores=ds)
    gref = gref.numpy()
    np.testing.assert_allclose(g, gref, rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(gref).max())))


@pytest.mark.parametrize("algo,B,L", [("softmax", 37, 10), ("softmax", 5, 100), ("dla", 33, 20), ("pairdebias", 18, 50),
                                      ("lambdarank", 18, 50), ("lambdarank", 3, 70), ("pairdebias", 2, 130)])
def test_oracle_losses(algo, B, L):
    from oracle import ultr_oracle as O
    rng = np.random.RandomState(B * 131 + L)
    scores = rng.normal(size=(B, L)).astype(np.float32)
    labels_LB = (rng.uniform(size=(L, B)) < 0.3).astype(np.float32)
    if algo == "lambdarank":
        labels_LB = rng.randint(0, 5, size=(L, B)).astype(np.float32)
    labels_LB[0, :] = np.maximum(labels_LB[0, :], 1.0)
    run = HipRun(8, [4], B, L, algo=algo)
    run.set_inputs(np.zeros((1, 8), np.float32), np.zeros((L, B), np.int32), labels_LB)
    s = torch.tensor(scores, requires_grad=True)
    y = torch.tensor(labels_LB.T.copy())
    tp = torch.tensor(rng.uniform(0.8, 1.2, size=L).astype(np.float32))
    tm = torch.tensor(rng.uniform(0.8, 1.2, size=L).astype(np.float32))
    if algo == "softmax":
        ipw = rng.uniform(1, 10, size=40)
        pw = O.ipw_weights(labels_LB, ipw)
        loss = O.softmax_loss(s, y, pw)
        ds, tail = run.loss(ipw_table=ipw, scores=scores)
        gs, hl = 1.0 / tail[1], tail[0] / tail[1]
    elif algo == "dla":
        q = torch.tensor(rng.normal(scale=0.3, size=L + 1).astype(np.float32), requires_grad=True)
        prop = O.denoising_net(q, B, L)
        with torch.no_grad():
            pw = O.normalized_weights(torch.softmax(prop, -1))
            rw = O.normalized_weights(torch.softmax(s, -1))
        rank = O.softmax_loss(s, y, pw)
        exam = O.softmax_loss(prop, y, rw)
        (gprop,) = torch.autograd.grad(exam, prop, retain_graph=True)
        loss = rank
        ds, tail = run.loss(aux=q.detach().numpy(), scores=scores)
        gs, hl = 1.0 / tail[1], tail[0] / tail[1]
        assert abs(tail[2] / tail[3] - float(exam)) < 1e-5
        np.testing.assert_allclose(tail[4:4 + L] / tail[3], gprop.sum(0).numpy(), rtol=1e-4, atol=1e-6)
    elif algo == "pairdebias":
        loss, PL, tpl, tml = O.pairdebias_loss(s, torch.tensor(labels_LB), tp, tm)
        ds, tail = run.loss(aux=np.concatenate([tp.numpy(), tm.numpy()]), scores=scores)
        gs, hl = 1.0, tail[0]
        np.testing.assert_allclose(tail[4:4 + L], tpl.detach().numpy(), rtol=2e-5, atol=1e-3)
        np.testing.assert_allclose(tail[4 + L:4 + 2 * L], tml.detach().numpy(), rtol=2e-5, atol=1e-3)
    else:
        loss, PL, tpl, tml = O.lambdarank_loss(s, y, tp, tm, 1.0)
        ds, tail = run.loss(aux=np.concatenate([tp.numpy(), tm.numpy()]), scores=scores)
        gs, hl = 1.0 / tail[1], tail[0] / tail[1]
        np.testing.assert_allclose(tail[4:4 + L] / tail[1], tpl.detach().numpy(), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(tail[4 + L:4 + 2 * L] / tail[1], tml.detach().numpy(), rtol=2e-5, atol=1e-6)
    (g,) = torch.autograd.grad(loss, s)
    assert abs(hl - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
    g = g.numpy()
    np.testing.assert_allclose(ds * gs, g, rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(g).max())))


def test_determinism_bitwise():
    """Same inputs twice -> bit-identical gradients and parameters (no atomics anywhere)."""
    d, m = load_golden("ipw_cfg2")
    outs = []
    for _ in range(2):
        run = make_run(m, "ipw_cfg2")
        run.set_inputs(d["s0_features"], d["s0_docids"], d["s0_labels"])
        run.forward(d["s0_pre_params"])
        run.loss(ipw_table=d["ipw_list"])
        g, _ = run.backward()
        params, state, _, sc = run.update(d["s0_pre_adagrad"])
        outs.append((g.copy(), params.copy(), state.copy(), sc.copy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name", ["ipw_tiny", "na_tiny", "ipw_cfg2", "ipw_odd"])
def test_fused_softmax_backward(name):
    """ultr_dnn_backward_softmax (loss fused into the backward prologue) == ultr_softmax_ce + ultr_dnn_backward:
    identical dscores and gradients (bitwise), loss / normaliser sums equal up to summation order."""
    from ultra_pytorch_amd import hip_ops
    d, m = load_golden(name)
    run = make_run(m, name)
    run.set_inputs(d["s0_features"], d["s0_docids"], d["s0_labels"])
    run.forward(d["s0_pre_params"])
    ipw = d["ipw_list"] if m["algo"] == "ipw" else None
    ds_ref, tail_ref = run.loss(ipw_table=ipw)
    g_ref, _ = run.backward()
    eng = run.eng
    eng.dscores.zero_()
    eng.grads.zero_()
    hip_ops.dnn_backward_softmax(run.shape, run.params, run.features, run.n_docs, run.docids, run.B, run.L, eng.saved, eng.scores,
                                 run.labels, eng.loss_ws, eng.bwd_ws, eng.grads, ipw_table=run.ipw, dscores_out=eng.dscores)
    torch.cuda.synchronize()
    assert np.array_equal(eng.dscores.cpu().numpy(), ds_ref)
    g = eng.grads.cpu().numpy()
    P = run.shape.n_params
    assert np.array_equal(g[:P], g_ref)
    np.testing.assert_allclose(g[P:P + 2], tail_ref[:2], rtol=1e-6)
    assert np.all(g[P + 2:] == 0)


@pytest.mark.parametrize("B,L,F,hidden,n_pad", [(256, 10, 136, [256, 256], 0), (37, 7, 24, [16, 8], 2), (64, 16, 136, [512, 256, 128], 0),
                                                (50, 5, 136, [32, 16], 1), (9, 1, 8, [8], 0), (33, 12, 260, [264, 12], 3)])
def test_fused_forward_backward_step(B, L, F, hidden, n_pad, monkeypatch):
    """ultr_train_step picks ONE fused forward+loss+backward launch for small NA/IPW batches (list_size <= 16): it must
    produce what the separate kernels produce (scores, loss, gradient, updated parameters) - and what the oracle says."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops, synthetic
    from ultra_pytorch_amd.ranking_model import init_flat_params
    shape = hip_ops.DnnShape(F, hidden, "elu")
    rng = np.random.RandomState(5)
    feats, ids, y = synthetic.make_batch(rng, B, L, F, n_pad=n_pad)  # PAD documents (id == n_docs) at the list tails
    ipw = np.asarray(synthetic.load_ipw(), np.float32)
    p0 = init_flat_params(shape, seed=3).numpy()
    out = {}
    for mode in ("fused", "separate"):
        monkeypatch.setenv("ULTR_NO_FUSED_FB", "0" if mode == "fused" else "1")
        eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax", learning_rate=0.05, max_gradient_norm=5.0)
        params, state = dev(p0.copy()), dev(np.zeros_like(p0))
        sc = eng.train_step(params, state, dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y), ipw_table=dev(ipw))
        torch.cuda.synchronize()
        out[mode] = dict(scores=eng.scores.cpu().numpy().copy(), grads=eng.grads.cpu().numpy().copy(),
                         params=params.cpu().numpy().copy(), scalars=sc.cpu().numpy().copy())
    f, s = out["fused"], out["separate"]
    np.testing.assert_allclose(f["scores"], s["scores"], atol=2e-6, rtol=1e-6)
    gmax = float(np.abs(s["grads"][: shape.n_params]).max())
    np.testing.assert_allclose(f["grads"], s["grads"], rtol=1e-5, atol=1e-6 * max(1.0, gmax))
    np.testing.assert_allclose(f["scalars"][:4], s["scalars"][:4], rtol=1e-5, atol=1e-6)
    r = O.train_step_softmax(p0, np.zeros_like(p0), F, hidden, feats, ids, y, ipw_list=ipw, lr=0.05, max_norm=5.0)
    np.testing.assert_allclose(f["scores"], r["scores"], atol=1e-5)
    assert abs(float(f["scalars"][0]) - r["loss"]) <= 1e-5 * max(1.0, abs(r["loss"]))
    gs = 1.0 / float(f["scalars"][3])
    np.testing.assert_allclose(f["grads"][: shape.n_params] * gs, r["grads"], rtol=1e-5,
                               atol=1e-6 * max(1.0, float(np.abs(r["grads"]).max())))


@pytest.mark.parametrize("fused", [True, False])
def test_ipw_table_saturates_at_last_entry(fused, monkeypatch):
    """SURVEY Appendix A.4 on the device: positions past the end of the IPW table reuse its LAST entry, unclicked
    documents weigh 0 - in the stand-alone loss / separate kernels and in the fused forward+loss+backward kernel."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops, synthetic
    from ultra_pytorch_amd.ranking_model import init_flat_params
    monkeypatch.setenv("ULTR_NO_FUSED_FB", "0" if fused else "1")
    B, L, F, hidden = 12, 10, 16, [8]
    shape = hip_ops.DnnShape(F, hidden, "elu")
    feats, ids, y = synthetic.make_batch(np.random.RandomState(3), B, L, F)
    ipw = np.asarray([1.0, 2.5, 7.0], np.float32)  # 3 entries for 10 positions
    p0 = init_flat_params(shape, seed=4).numpy()
    eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax")
    params, state = dev(p0.copy()), dev(np.zeros_like(p0))
    sc = eng.train_step(params, state, dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y), ipw_table=dev(ipw))
    torch.cuda.synchronize()
    r = O.train_step_softmax(p0, np.zeros_like(p0), F, hidden, feats, ids, y, ipw_list=ipw, lr=0.05, max_norm=5.0)
    assert abs(float(sc[0]) - r["loss"]) <= 1e-5 * max(1.0, abs(r["loss"]))
    gs = 1.0 / float(sc[3])
    np.testing.assert_allclose(eng.grads.cpu().numpy()[: shape.n_params] * gs, r["grads"], rtol=1e-5,
                               atol=1e-6 * max(1.0, float(np.abs(r["grads"]).max())))
