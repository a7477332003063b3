"""dnn_fwdw_kernel / dnn_bwdw_kernel - the wide-tile forward and row-local backward (17 .. 48 rows per workgroup behind one
split-half weight stream, round 5).  The forward against the
oracle's DNN forward (DNN.py:41-88, base_algorithm.py:118-154): scores at 1e-5, and through the saved activations / statistics it
leaves for the backward, the gradients of a whole softmax step at the bar of tests/test_gpu_full_size.py.  Shapes pick every
code path: two and three MFMA row tiles, a ragged last workgroup, chunks x slices of the contraction (fewer than sixteen
32-column chunks, odd slice counts), more chunks than waves, one / two / three float4 per lane in the LayerNorms, PAD documents,
evaluation (nothing saved), every activation.  The backward runs inside ultr_train_step (it needs the weight copies): one whole step
(IPW softmax, whose loss then runs as its own launch, and DLA) against the oracle's step (ipw_rank.py:102-182, dla.py:179-266)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.hipref import HipRun  # noqa: E402
from tests import margins  # noqa: E402



@pytest.fixture(autouse=True)
def wide_whenever_legal(monkeypatch):
    """ULTR_FWD_WIDE=2: the wide-tile forward wherever it is legal (the default takes it by a measured rule, which sends some of the
    small shapes below to the 16-row kernel)"""
    from ultra_pytorch_amd import _lib
    monkeypatch.setenv("ULTR_FWD_WIDE", "2")
    _lib.load().ultr_config_reload()
    yield
    monkeypatch.undo()
    _lib.load().ultr_config_reload()


SHAPES = {
    # name: (F, hidden, B, L, act, n_pad)
    "cfg3_three_tiles": (136, [512, 256, 128], 512, 20, "elu", 0),        # 40 rows x 256 workgroups
    "two_tiles_ragged": (136, [256, 256], 300, 23, "elu", 2),             # 27 rows, the last workgroup holds 15
    "cfg4_wide_input": (700, [512, 256, 128], 256, 50, "elu", 0),         # 25 rows x 512 workgroups, three float4 per lane
    "odd_slices": (220, [96, 64], 200, 40, "relu", 3),                    # 3 and 2 chunks: 4 and 3 slices of the contraction
    "chunks_beyond_waves": (136, [768, 32], 256, 40, "tanh", 0),          # 24 chunks on 16 waves; then ONE chunk in 12 slices
    "sigmoid_narrow": (48, [64, 64, 32], 128, 60, "sigmoid", 5),
    "one_hidden_layer": (136, [256], 400, 30, "elu", 1),                  # backward: the scorer's row pass only, no product
    "four_hidden_layers": (64, [128, 256, 128, 64], 350, 25, "elu", 0),
}


def tile_rows(F, hidden, act, n_rows, training=True):
    from ultra_pytorch_amd import _lib, hip_ops
    shape = hip_ops.DnnShape(F, hidden, act)
    return _lib.load().ultr_dnn_forward_tile_rows(shape.desc, n_rows, 1 if training else 0)


def inputs(name):
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import synthetic
    F, hidden, B, L, act, n_pad = SHAPES[name]
    rng = np.random.RandomState(11)
    feats, ids, y = synthetic.make_batch(rng, B, L, F, clicks=True, n_pad=n_pad)
    params = O.init_params(F, hidden, seed=5)
    # LayerNorm affine parameters away from their (1, 0) initial values, a few weights large
    lay = O.param_layout(F, hidden)
    for n, s, o in lay:
        k = int(np.prod(s))
        if "layer_norm" in n:
            params[o:o + k] += rng.normal(scale=0.2, size=k).astype(np.float32)
    return feats, ids, y, params


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("train", [True, False], ids=["train", "eval"])
def test_wide_forward_scores_match_oracle(name, train):
    from oracle import ultr_oracle as O
    F, hidden, B, L, act, n_pad = SHAPES[name]
    R = tile_rows(F, hidden, act, B * L, train)
    assert R > 1000, "shape does not take the wide-tile kernel (ultr_dnn_forward_tile_rows = %d)" % R
    R -= 1000
    feats, ids, y, params = inputs(name)
    run = HipRun(F, hidden, B, L, algo="softmax", act=act)
    run.set_inputs(feats, ids, y)
    scores = run.forward(params, train=train)
    ref = O.ranking_scores(torch.from_numpy(params), F, hidden, feats, ids, act).numpy()
    print("%s: %d rows per workgroup, max |score diff| %.2e" % (name, R, np.abs(scores - ref).max()))
    margins.check("wide_tiles/%s_%s" % (name, "train" if train else "eval"), "scores_max_abs_diff", np.abs(scores - ref).max())
    np.testing.assert_allclose(scores, ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("name", list(SHAPES))
def test_wide_forward_feeds_the_backward(name):
    """forward (wide tiles) -> softmax loss -> backward: the gradients come out of what the forward saved (xhat_0, the
    activations, the LayerNorm statistics of every layer)"""
    from oracle import ultr_oracle as O
    F, hidden, B, L, act, n_pad = SHAPES[name]
    assert tile_rows(F, hidden, act, B * L) > 1000
    feats, ids, y, params = inputs(name)
    run = HipRun(F, hidden, B, L, algo="softmax", act=act)
    run.set_inputs(feats, ids, y)
    run.forward(params)
    ds, tail = run.loss()
    g, tail2 = run.backward()
    gs = 1.0 / tail2[1]
    x = O.gather_rows(feats, ids).numpy()
    dsc = (ds * gs).T.reshape(-1)
    gref = O.dnn_backward_manual(params, F, hidden, x, dsc, act)
    terms = O.dnn_backward_manual(params, F, hidden, x, dsc, act, abs_terms=True)
    d = np.abs(g * gs - gref)
    print("%s: max |g - g_ref| / (|g_ref| + terms) = %.2e" % (name, (d / np.maximum(np.abs(gref) + terms, 1e-30)).max()))
    margins.check("wide_tiles/%s_backward" % name, "grads_max_diff_over_abs_g_plus_terms", (d / np.maximum(np.abs(gref) + terms, 1e-30)).max())
    assert (d <= 1e-5 * (np.abs(gref) + terms)).all()


def test_wide_forward_is_deterministic_and_knob_switches_it_off(monkeypatch):
    from ultra_pytorch_amd import _lib
    F, hidden, B, L, act, n_pad = SHAPES["cfg3_three_tiles"]
    feats, ids, y, params = inputs("cfg3_three_tiles")
    run = HipRun(F, hidden, B, L, algo="softmax", act=act)
    run.set_inputs(feats, ids, y)
    a = run.forward(params)
    b = run.forward(params)
    assert np.array_equal(a, b)
    monkeypatch.setenv("ULTR_FWD_WIDE", "0")
    _lib.load().ultr_config_reload()
    assert tile_rows(F, hidden, act, B * L) == 16
    c = run.forward(params)
    # the 16-row kernel sums the same products in another order (no slices of the contraction, other tile shapes)
    np.testing.assert_allclose(a, c, atol=2e-6)


def backward_tile_rows(F, hidden, act, n_rows):
    from ultra_pytorch_amd import _lib, hip_ops
    return _lib.load().ultr_dnn_backward_tile_rows(hip_ops.DnnShape(F, hidden, act).desc, n_rows)


@pytest.mark.parametrize("name", [k for k in SHAPES if k != "chunks_beyond_waves"])  # (a 768-wide LayerNorm: the per-layer backward)
@pytest.mark.parametrize("algo", ["softmax", "dla"])
def test_wide_backward_step_matches_oracle(name, algo):
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops, synthetic
    from tests.hipref import dev
    F, hidden, B, L, act, n_pad = SHAPES[name]
    R = backward_tile_rows(F, hidden, act, B * L)
    assert R > 1000, "shape does not take the wide-tile backward (ultr_dnn_backward_tile_rows = %d)" % R
    feats, ids, y, params = inputs(name)
    rng = np.random.RandomState(3)
    lr = 0.05
    if algo == "softmax":
        aux = None
        ref = O.train_step_softmax(params, np.zeros_like(params), F, hidden, feats, ids, y, ipw_list=synthetic.load_ipw(), lr=lr, act=act)
    else:
        aux = rng.normal(scale=0.2, size=L + 1).astype(np.float32)
        ref = O.dla_step(params, aux, F, hidden, feats, ids, y, lr=lr, act=act)
    shape = hip_ops.DnnShape(F, hidden, act)
    eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo=algo, learning_rate=lr)
    p = dev(params)
    st = None if algo == "dla" else dev(np.zeros_like(params))
    a = None if aux is None else dev(aux)
    tab = dev(np.asarray(synthetic.load_ipw(), np.float32)) if algo == "softmax" else None
    eng.train_step(p, st, dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y), aux=a, ipw_table=tab)
    sc = eng.read_scalars()
    scores = eng.scores.cpu().numpy()
    np.testing.assert_allclose(scores, ref["scores"], atol=1e-5, rtol=1e-5)
    assert abs(sc[0] - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
    n = shape.n_params
    tail = eng.grads[n:].cpu().numpy()
    gs = 1.0 / tail[1]
    g, gref = eng.grads[:n].cpu().numpy() * gs, ref["grads"]
    d = np.abs(g - gref)
    print("%s / %s: %d rows per workgroup, max |g - g_ref| / max|g_ref| = %.2e" % (name, algo, R - 1000, d.max() / np.abs(gref).max()))
    np.testing.assert_allclose(g, gref, rtol=1e-5, atol=1e-5 * float(np.abs(gref).max()))  # (the bar of tests/test_gpu_knobs.py)
    assert abs(sc[1] - ref["norm"]) <= 1e-5 * ref["norm"]
    eng.close()
