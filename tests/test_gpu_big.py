"""The per-layer big-batch path of the DNN training step (csrc/ultr_dnn_big.hip: row statistics -> tiled GEMM with the
LayerNorm applied by the A-operand producer -> bias/activation epilogue; LayerNorm-backward row kernels + tiled dgrad
GEMMs) against the oracle and against the row-tile kernels it replaces for big batches.  Forced on small shapes with
ULTR_BIG_FWD / ULTR_BIG_BWD = 2 so the oracle finishes in seconds; the size rule itself is checked at config 4's shape
in tests/test_gpu_full_size.py.  Tolerances: scores 1e-5, gradients 2e-5 rel + 2e-6*max|g| (as test_gpu_parity)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.hipref import HipRun, dev  # noqa: E402
from tests.test_gpu_parity import synth  # noqa: E402

# F, hidden, B, L, act: all widths multiples of 4 (the path's legality rule); rows not a multiple of the 32-row blocks;
# one hidden layer (top kernel only), 1024-wide rows (four 16-byte chunks per lane), config 4's and config 2's stacks
SHAPES = [(136, [512, 256, 128], 64, 10, "elu"), (700, [512, 256, 128], 37, 9, "elu"), (24, [16], 33, 7, "elu"),
          (136, [256, 256], 50, 10, "relu"), (1024, [1024, 64], 9, 11, "elu"), (44, [20, 12, 8, 4], 21, 5, "elu")]


def perturbed_params(O, F, hidden, rng):
    params = O.init_params(F, hidden, seed=3)
    for name, shape, off in O.param_layout(F, hidden):
        if "layer_norm" in name:
            n = int(np.prod(shape))
            params[off:off + n] += rng.uniform(-0.3, 0.3, size=n).astype(np.float32)
    return params


@pytest.mark.parametrize("F,hidden,B,L,act", SHAPES)
@pytest.mark.parametrize("which", ["both", "fwd", "bwd"])
def test_big_path_vs_oracle(F, hidden, B, L, act, which, monkeypatch):
    from oracle import ultr_oracle as O
    if max([F] + hidden) > 512 + 256 and which != "both":
        pytest.skip("the row-tile kernels do not hold 1024-wide rows in LDS: only the per-layer path runs this shape")
    monkeypatch.setenv("ULTR_BIG_FWD", "2" if which in ("both", "fwd") else "0")
    monkeypatch.setenv("ULTR_BIG_BWD", "2" if which in ("both", "bwd") else "0")
    feats, ids, labels = synth(F, B, L, 5)
    rng = np.random.RandomState(9)
    params = perturbed_params(O, F, hidden, rng)
    run = HipRun(F, hidden, B, L, act=act)
    run.set_inputs(feats, ids, labels)
    scores = run.forward(params)
    p = torch.tensor(params, requires_grad=True)
    ref = O.ranking_scores(p, F, hidden, feats, ids, act=act)
    np.testing.assert_allclose(scores, ref.detach().numpy(), atol=1e-5, rtol=1e-5)
    ds = rng.normal(size=(B, L)).astype(np.float32)
    (gref,) = torch.autograd.grad((ref * torch.tensor(ds)).sum(), p)
    g, _ = run.backward(dscores=ds)
    gref = gref.numpy()
    np.testing.assert_allclose(g, gref, rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(gref).max())))


@pytest.mark.parametrize("algo", ["softmax", "dla", "pairdebias", "lambdarank"])
def test_big_path_train_steps_match_row_tile_path(algo, monkeypatch):
    """Whole training steps (forward, loss, backward, clip, Adagrad) through ultr_train_step: the per-layer path and the
    row-tile kernels must agree (NA/IPW must take the stand-alone loss stage correctly).  Step 1: scores, gradient and
    updated parameters; steps 1-3: the loss.  Parameters whose gradient is rounding noise (the scorer's bias under a
    softmax loss: sum_l dscores = 0 exactly) are excluded from the parameter check - the first Adagrad step moves them by
    lr * sign(noise) in BOTH paths, as in the reference."""
    from ultra_pytorch_amd import engine, hip_ops, synthetic
    from ultra_pytorch_amd.ranking_model import init_flat_params
    F, hidden, B, L = 136, [512, 256, 128], 48, 10
    shape = hip_ops.DnnShape(F, hidden, "elu")
    rng = np.random.RandomState(11)
    feats, ids, y = synthetic.make_batch(rng, B, L, F, n_pad=2)
    ipw = np.asarray(synthetic.load_ipw(), np.float32)
    p0 = init_flat_params(shape, seed=3).numpy()
    out = {}
    for mode in ("big", "tile"):
        monkeypatch.setenv("ULTR_BIG_FWD", "2" if mode == "big" else "0")
        monkeypatch.setenv("ULTR_BIG_BWD", "2" if mode == "big" else "0")
        monkeypatch.setenv("ULTR_NO_FUSED_FB", "1")
        eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo=algo, learning_rate=0.05, max_gradient_norm=5.0)
        params, state = dev(p0.copy()), (None if algo == "dla" else dev(np.zeros_like(p0)))
        arng = np.random.RandomState(4)
        kw = {}
        if algo == "softmax":
            kw = dict(ipw_table=dev(ipw))
        elif algo == "dla":  # the DenoisingNet's Linear(L, 1): [W | bias]
            kw = dict(aux=dev((0.1 * arng.randn(L + 1)).astype(np.float32)))
        else:  # t_plus | t_minus
            kw = dict(aux=dev(np.linspace(0.9, 1.2, 2 * L).astype(np.float32)))
        losses, first = [], None
        for it in range(3):
            sc = eng.train_step(params, state, dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y), **kw)
            losses.append(float(sc[0].item()))
            if it == 0:
                first = dict(scores=eng.scores.cpu().numpy().copy(), grads=eng.grads.cpu().numpy()[: shape.n_params].copy(),
                             params=params.cpu().numpy().copy())
        out[mode] = dict(first, losses=np.asarray(losses))
    a, b = out["big"], out["tile"]
    np.testing.assert_allclose(a["scores"], b["scores"], atol=1e-5, rtol=1e-5)
    gmax = float(np.abs(b["grads"]).max())
    np.testing.assert_allclose(a["grads"], b["grads"], rtol=2e-5, atol=2e-6 * max(1.0, gmax))
    solid = np.abs(b["grads"]) > 1e-4 * gmax
    np.testing.assert_allclose(a["params"][solid], b["params"][solid], rtol=1e-5, atol=2e-6)
    n = 1 if algo == "dla" else 3  # DLA's per-step optimizers update by lr * sign(g): noise-level gradients fork the paths
    np.testing.assert_allclose(a["losses"][:n], b["losses"][:n], rtol=2e-5, atol=1e-6)


def test_xhat0_marker_follows_the_forward_that_ran(monkeypatch):
    """A per-layer forward over a wide gathered input (K_0 > 512) leaves xhat_0 in saved.x_0 and says so in the marker word
    behind the saved activations; the weight-gradient launch then contracts layer 0 with it instead of gathering and
    normalising again.  Every forward writes the marker: switch the SAME engine (same `saved` buffer) between the two
    forwards and the gradients must stay right both ways."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import _lib
    F, hidden, B, L = 700, [512, 256, 128], 37, 9
    feats, ids, labels = synth(F, B, L, 5)
    rng = np.random.RandomState(9)
    params = perturbed_params(O, F, hidden, rng)
    ds = rng.normal(size=(B, L)).astype(np.float32)
    p = torch.tensor(params, requires_grad=True)
    ref = O.ranking_scores(p, F, hidden, feats, ids)
    (gref,) = torch.autograd.grad((ref * torch.tensor(ds)).sum(), p)
    gref = gref.numpy()
    monkeypatch.setenv("ULTR_BIG_BWD", "0")
    monkeypatch.setenv("ULTR_BIG_FWD", "2")
    run = HipRun(F, hidden, B, L)
    run.set_inputs(feats, ids, labels)
    for mode in ("2", "0", "2", "0"):
        monkeypatch.setenv("ULTR_BIG_FWD", mode)
        _lib.load().ultr_config_reload()
        scores = run.forward(params)
        np.testing.assert_allclose(scores, ref.detach().numpy(), atol=1e-5, rtol=1e-5)
        g, _ = run.backward(dscores=ds)
        np.testing.assert_allclose(g, gref, rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(gref).max())), err_msg="forward mode " + mode)


def test_big_path_is_deterministic(monkeypatch):
    """Fixed-order partial sums everywhere: two runs give bit-identical gradients."""
    from oracle import ultr_oracle as O
    monkeypatch.setenv("ULTR_BIG_FWD", "2")
    monkeypatch.setenv("ULTR_BIG_BWD", "2")
    F, hidden, B, L = 136, [512, 256, 128], 64, 10
    feats, ids, labels = synth(F, B, L, 5)
    params = O.init_params(F, hidden, seed=3)
    ds = np.random.RandomState(2).normal(size=(B, L)).astype(np.float32)
    got = []
    for _ in range(2):
        run = HipRun(F, hidden, B, L)
        run.set_inputs(feats, ids, labels)
        s = run.forward(params).copy()
        g, _ = run.backward(dscores=ds)
        got.append((s, g.copy()))
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])


def test_wide_layer_small_batch_softmax_train_step():
    """A layer wider than the row-tile kernels hold (1024 > 512) at a SMALL batch under NA / IPW: ultr_train_step must route
    the backward through the stand-alone loss + per-layer path by itself (ultr_dnn_backward_softmax used to return
    ULTR_E_UNSUPPORTED here, while the other algorithms took that route) - and match the oracle."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops
    F, hidden, B, L = 64, [1024, 32], 9, 7
    feats, ids, labels = synth(F, B, L, 8)
    params = O.init_params(F, hidden, seed=4)
    ipw = np.linspace(1.0, 4.0, 5)
    ref = O.train_step_softmax(params, np.zeros_like(params), F, hidden, feats, ids, labels, ipw_list=ipw)
    shape = hip_ops.DnnShape(F, hidden, "elu")
    eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax")
    p, st = dev(params), dev(np.zeros_like(params))
    eng.train_step(p, st, dev(feats), feats.shape[0], dev(ids, torch.int32), dev(labels), ipw_table=dev(ipw.astype(np.float32)))
    sc = eng.read_scalars()
    assert abs(sc[0] - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
    np.testing.assert_allclose(eng.scores.cpu().numpy(), ref["scores"], atol=1e-5)
    g = eng.grads[:shape.n_params].cpu().numpy() / sc[3]
    np.testing.assert_allclose(g, ref["grads"], rtol=2e-5, atol=2e-6 * float(np.abs(ref["grads"]).max()))
