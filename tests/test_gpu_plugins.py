"""The plugin classes (ultra_pytorch_amd.learning_algorithm.* / ranking_model.*) driven exactly the way the
reference's main.py drives ultra.learning_algorithm.*: class paths from a settings dict, numpy `input_feed`
dicts in the feed's layout, train()/validation() return triples.  Checked against the golden vectors."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.hipref import load_golden  # noqa: E402

CLS = {"na": "NavieAlgorithm", "ipw": "IPWrank", "dla": "DLA", "pairdebias": "PairDebias", "lambdarank": "LambdaRank",
       "regem": "RegressionEM"}


class DataSet:
    def __init__(self, feature_size):
        self.feature_size = feature_size


def make_feed(algo, feats, docids, labels):
    L = docids.shape[0]
    feed = {algo.letor_features_name: feats.astype(np.float64)}  # the reference feed hands over Python floats (f64)
    for l in range(L):
        feed[algo.docid_inputs_name[l]] = docids[l].astype(np.float32)  # ids arrive as float32 (feed quirk)
        feed[algo.labels_name[l]] = labels[l].astype(np.float32)
    return feed


def build(m, model_cls="DNN", extra=None):
    from ultra_pytorch_amd.utils import find_class
    hidden = m["hidden"]
    exp = {
        "learning_algorithm": "ultra_pytorch_amd.learning_algorithm." + CLS[m["algo"]],
        "learning_algorithm_hparams": m.get("algo_hparams", ""),
        "ranking_model": "ultra_pytorch_amd.ranking_model." + model_cls,
        "ranking_model_hparams": (("hidden_layer_sizes=%s" % json.dumps(hidden)) if hidden is not None else "") + m.get("model_extra", ""),
        "max_candidate_num": m["L"], "selection_bias_cutoff": m["L"],
        "metrics": ["ndcg", "mrr", "err"], "metrics_topn": [1, 3, 5, 10],
    }
    if extra:
        exp.update(extra)
    return find_class(exp["learning_algorithm"])(DataSet(m["F"]), exp)


def load_flat(model, flat):
    sd = {}
    for name, shape, off in model.shape.layout():
        sd[name] = torch.from_numpy(flat[off:off + int(np.prod(shape))].reshape(shape).copy())
    model.load_state_dict(sd)


@pytest.mark.parametrize("name", ["na_tiny", "ipw_tiny", "dla_tiny", "pairdebias_tiny", "lambdarank_tiny", "na_linear", "ipw_cfg2",
                                  "regem_tiny", "na_tanh", "na_sigmoid", "ipw_l2", "na_l2", "dla_l2", "pairdebias_l2", "regem_l2"])
def test_train_matches_golden(name):
    d, m = load_golden(name)
    for a in ("tanh", "sigmoid"):  # activation_func travels in ranking_model_hparams (DNN.py:25-32)
        if name.endswith("_" + a):
            m = dict(m, model_extra=",activation_func=" + a)
    algo = build(m, model_cls="Linear" if m["model"] == "Linear" else "DNN")
    assert list(algo.model.state_dict().keys()) == m["param_keys"]  # checkpoint interchange
    L = m["L"]
    for t in range(m["n_steps"]):
        p = "s%d_" % t
        load_flat(algo.model, d[p + "pre_params"])  # teacher forcing: every step starts from the reference's state
        if m["algo"] != "dla":
            algo.state_sum.copy_(torch.from_numpy(d[p + "pre_adagrad"]))
        if m["algo"] == "dla":
            algo.propensity_model.flat_params.copy_(torch.from_numpy(d[p + "pre_prop_params"]))
        if m["algo"] in ("pairdebias", "lambdarank"):
            algo.t_state.copy_(torch.from_numpy(np.concatenate([d[p + "pre_t_plus"].ravel(), d[p + "pre_t_minus"].ravel()])))
        if m["algo"] == "regem":
            algo.propensity_state.copy_(torch.from_numpy(d[p + "pre_propensity"].ravel()))
            algo.uniforms = torch.from_numpy(d[p + "uniforms"]).cuda()  # the reference's recorded Bernoulli draw
        feed = make_feed(algo, d[p + "features"], d[p + "docids"], d[p + "labels"])
        loss, out, summary = algo.train(feed)
        ref = float(d[p + "loss"])
        assert out is None and isinstance(summary, dict)
        assert abs(loss - ref) <= 1e-5 * max(1.0, abs(ref))
        g = d[p + "grads"]
        sel = np.abs(g) > 1e-6 * max(1.0, float(np.abs(g).max()))
        got = algo.model.flat_params.cpu().numpy()
        np.testing.assert_allclose(got[sel], d[p + "post_params"][sel], atol=5e-6, rtol=1e-5)
        if m["algo"] in ("pairdebias", "lambdarank"):
            np.testing.assert_allclose(algo.t_plus.cpu().numpy(), d[p + "post_t_plus"], atol=1e-6)
            np.testing.assert_allclose(algo.t_minus.cpu().numpy(), d[p + "post_t_minus"], atol=1e-6)
        if m["algo"] == "regem":
            np.testing.assert_allclose(algo.propensity.cpu().numpy(), d[p + "post_propensity"], atol=1e-6)
        if m["algo"] == "dla":
            np.testing.assert_allclose(algo.propensity_model.flat_params.cpu().numpy(), d[p + "post_prop_params"], atol=1e-6)
        if m["algo"] == "ipw":
            np.testing.assert_allclose(np.asarray([feed["propensity_weights%d" % l] for l in range(L)]).T, d[p + "pw"], rtol=1e-6)
    assert algo.global_step == m["n_steps"]


@pytest.mark.parametrize("name", ["valid_tiny", "valid_odd"])
def test_validation_matches_golden(name):
    from ultra_pytorch_amd.utils import metrics
    d, m = load_golden(name)
    metrics.RankingMetricKey.MAX_LABEL = m["max_label"]  # global set by the data loader (data_utils.py:96), as in the reference
    m2 = dict(m, algo="ipw")
    algo = build(m2, extra={"selection_bias_cutoff": min(10, m["L"])})
    load_flat(algo.model, d["params"])
    for b in range(int(d["n_batches"])):
        p = "b%d_" % b
        feed = make_feed(algo, d[p + "features"], d[p + "docids"], d[p + "labels"])
        loss, scores, summary = algo.validation(feed)
        assert loss is None
        np.testing.assert_allclose(scores.cpu().numpy(), d[p + "scores"], atol=1e-5)  # unmasked scores
        for metric in ("ndcg", "mrr", "err"):
            for n in m["topn"]:
                assert abs(summary["%s_%d" % (metric, n)] - float(d[p + "metric_%s_%d" % (metric, n)])) < 1e-6, (metric, n)


def test_dnn_build_contract():
    """ranking_model.DNN.build(list of L [B,F] tensors) -> sequence of L [B,1] tensors (DNN.py:58-88)."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd.ranking_model import DNN
    F, hidden, B, L = 24, [16, 8], 6, 5
    model = DNN("hidden_layer_sizes=[16,8]", F).cuda()
    rng = np.random.RandomState(0)
    xs = [torch.tensor(rng.uniform(-1, 1, size=(B, F)).astype(np.float32)) for _ in range(L)]
    outs = model.build([x.cuda() for x in xs])
    assert len(outs) == L and tuple(outs[0].shape) == (B, 1)
    ref = O.dnn_forward(model.flat_params.cpu(), F, hidden, torch.cat(xs, 0)).view(L, B)
    got = torch.cat([o.view(1, B) for o in outs], 0).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=1e-5)
    sd = model.state_dict()
    assert list(sd.keys())[0] == "sequential.layer_norm0.weight"


def test_unsupported_options_raise():
    d, m = load_golden("ipw_tiny")
    with pytest.raises(NotImplementedError):
        build(dict(m, algo_hparams="loss_func=sigmoid_loss"))
    with pytest.raises(TypeError):  # as the reference: ACT_FUNC_DIC['selu'] is a plain function, add_module rejects it
        build(dict(m, model_extra=",activation_func=selu"))


def test_lazy_propensity_columns_behave_like_the_reference_lists():
    """IPWrank.train leaves `propensity_weights{l}` in the caller's feed (ipw_rank.py:118-128) - materialised lazily here."""
    d, m = load_golden("ipw_tiny")
    algo = build(m)
    feed = make_feed(algo, d["s0_features"], d["s0_docids"], d["s0_labels"])
    algo.train(feed)
    col = feed["propensity_weights0"]
    assert len(col) == m["B"] and isinstance(col[0], float) and list(col) == col.tolist()
    np.testing.assert_allclose(np.asarray(col), d["s0_pw"][:, 0], rtol=1e-6)
    np.testing.assert_allclose(algo.propensity_weights, d["s0_pw"], rtol=1e-6)


def test_regression_em_device_rng():
    """Without injected uniforms the Bernoulli draw comes from the device Philox stream: pseudo-label rate must match
    the posterior mean, the draw must be reproducible for a fixed (seed, step) and differ between steps."""
    from ultra_pytorch_amd import hip_ops
    B, L = 512, 10
    rng = np.random.RandomState(0)
    scores = torch.from_numpy(rng.normal(size=(B, L)).astype(np.float32)).cuda()
    labels = torch.from_numpy((rng.uniform(size=(L, B)) < 0.2).astype(np.float32)).cuda()
    prop = torch.full((L,), 0.9, device="cuda")
    ds = torch.empty(B, L, device="cuda")
    ws = torch.zeros(hip_ops.loss_workspace_bytes(B, L) // 4, device="cuda")
    ys = []
    for step in (0, 0, 1):
        y = torch.empty(B, L, device="cuda")
        hip_ops.regem_loss(scores, labels, prop, B, L, ds, ws, seed=7, step=step, pseudo_out=y)
        ys.append(y.cpu().numpy())
    assert np.array_equal(ys[0], ys[1]) and not np.array_equal(ys[0], ys[2])
    gamma = 1.0 / (1.0 + np.exp(-scores.cpu().numpy().astype(np.float64)))
    c = labels.cpu().numpy().T
    p_r1 = c + (1 - c) * (0.1 * gamma / (1 - 0.9 * gamma))
    assert set(np.unique(ys[0])) <= {0.0, 1.0}
    assert abs(ys[0].mean() - p_r1.mean()) < 4 * np.sqrt(0.25 / (B * L))


def test_plugin_trains_through_a_weight_outside_the_split_half_range(monkeypatch):
    """IPWrank.train() the way main.py calls it, on a checkpoint whose hidden weight of 90 is outside what the split-half weight
    copies serve exactly-with-margin (|w| < 64; they overflow at 128): the reference trains any weight (base_algorithm.py:
    208-226), so the plugin must too - it switches to the fp32 matrix-core products with a warning and returns the reference's
    loss (ipw_cfg2 fixture, one planted weight, against the oracle)."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import hip_ops
    for k in hip_ops.H3_KNOBS:
        monkeypatch.setenv(k, "1")
    d, m = load_golden("ipw_cfg2")
    algo = build(m)
    flat = d["s0_pre_params"].copy()
    for name, shape, off in O.param_layout(m["F"], m["hidden"]):
        if name.endswith("linear1.weight"):
            flat[off + 7] = 90.0
    try:
        load_flat(algo.model, flat)
        feed = make_feed(algo, d["s0_features"], d["s0_docids"], d["s0_labels"])
        with pytest.warns(RuntimeWarning, match="fp32 matrix cores"):
            loss, _, _ = algo.train(feed)
        ref = O.train_step_softmax(flat, d["s0_pre_adagrad"], m["F"], m["hidden"], d["s0_features"], d["s0_docids"].astype(np.int32),
                                   d["s0_labels"], ipw_list=d["ipw_list"], lr=m["lr"], max_norm=m["max_gradient_norm"])
        assert abs(loss - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
        assert not hip_ops.split_half_enabled(algo.model.shape) and hip_ops.split_half_enabled()
        loss2, _, _ = algo.train(feed)  # and keeps training
        assert np.isfinite(loss2)
    finally:
        monkeypatch.undo()
        algo.model.shape.lib.ultr_config_reload()
