"""Edge cases of the hot path on the GPU against the oracle: degenerate shapes (one list, one document), lists
without any click under IPW (0/0 -> 0, base_algorithm.py:26-27), lists made of PAD documents only, an all-zero label
batch, every document of the batch being the same row, and the longest lists the list-wise kernels take."""
import os

import numpy as np
import pytest
from tests import margins
import torch

pytestmark = pytest.mark.gpu

from tests.hipref import HipRun, dev  # noqa: E402


def _step(F, hidden, B, L, feats, ids, labels, ipw, algo="softmax"):
    from oracle import ultr_oracle as O
    params = O.init_params(F, hidden, seed=4)
    run = HipRun(F, hidden, B, L, algo=algo, learning_rate=0.05, max_gradient_norm=5.0)
    run.set_inputs(feats, ids, labels)
    scores = run.forward(params)
    ds, tail = run.loss(ipw_table=ipw)
    g, tail2 = run.backward()
    p_new, s_new, _, sc = run.update(np.zeros_like(params))
    ref = O.train_step_softmax(params, np.zeros_like(params), F, hidden, feats, ids, labels, ipw_list=ipw, lr=0.05, max_norm=5.0)
    np.testing.assert_allclose(scores, ref["scores"], atol=1e-5, rtol=1e-5)
    assert np.isfinite(g).all() and np.isfinite(p_new).all()
    assert abs(float(sc[0]) - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
    gs = 1.0 / float(tail2[1])
    gmax = max(1.0, float(np.abs(ref["grads"]).max()))
    np.testing.assert_allclose(g * gs, ref["grads"], rtol=2e-5, atol=2e-6 * gmax)
    sel = np.abs(ref["grads"]) > 1e-5 * gmax  # the first Adagrad step is sign-like: compare where the sign is determined
    np.testing.assert_allclose(p_new[sel], ref["params"][sel], atol=5e-6, rtol=1e-5)
    return scores, g, sc


# (1500, 2) and (70000, 1): more than 1024 / 65536 loss partials (one per list) -> the two-level fold of the step tail
@pytest.mark.parametrize("B,L", [(1, 1), (1, 7), (2, 1), (300, 1), (1500, 2), (70000, 1)])
def test_degenerate_shapes(B, L):
    rng = np.random.RandomState(B * 10 + L)
    F, hidden = 12, [16, 8]
    feats = rng.uniform(-1, 1, size=(B * L, F)).astype(np.float32)
    ids = rng.permutation(B * L).astype(np.int32).reshape(L, B)
    labels = np.ones((L, B), np.float32)
    _step(F, hidden, B, L, feats, ids, labels, None)


@pytest.mark.parametrize("F,hidden,B,L", [(136, [256, 256], 256, 10), (20, [32], 9, 13)])  # fused kernel / separate kernels
def test_lists_without_clicks_under_ipw(F, hidden, B, L):
    """IPW weights are 0 on unclicked documents, so a list without a click has S_b = 0 and q = 0/0: the reference maps
    the NaN to 0 (base_algorithm.py:26-27) and the list contributes nothing - not a NaN - to loss and gradient."""
    rng = np.random.RandomState(3)
    feats = rng.uniform(-1, 1, size=(B * L, F)).astype(np.float32)
    ids = rng.permutation(B * L).astype(np.int32).reshape(L, B)
    labels = (rng.uniform(size=(L, B)) < 0.3).astype(np.float32)
    labels[:, ::3] = 0.0  # every third list: no click at all
    labels[0, 1] = 1.0
    ipw = np.linspace(1.0, 9.0, 40)
    _step(F, hidden, B, L, feats, ids, labels, ipw)


def test_all_pad_lists_and_shared_rows():
    """Lists made of PAD documents only (id == n_docs -> the zero feature row, base_algorithm.py:148-149) and a batch
    whose real documents are all the same row."""
    rng = np.random.RandomState(8)
    F, hidden, B, L = 24, [32, 16], 12, 10
    feats = rng.uniform(-1, 1, size=(5, F)).astype(np.float32)
    ids = np.full((L, B), 2, np.int32)   # everybody is document 2 ...
    ids[:, 4] = 5                        # ... list 4 is PAD only (n_docs = 5)
    ids[7:, 9] = 5                       # a ragged tail of PADs
    labels = (rng.uniform(size=(L, B)) < 0.4).astype(np.float32)
    labels[0, :] = 1.0
    scores, g, _ = _step(F, hidden, B, L, feats, ids, labels, None)
    assert np.allclose(scores[4], scores[4][0])  # one value: the score of the zero row
    from ultra_pytorch_amd import engine
    ev = engine.EvalEngine(HipRun(F, hidden, B, L).shape, B, L, torch.device("cuda"), topn=(1, 3, 10))
    from oracle import ultr_oracle as O
    params = torch.tensor(O.init_params(F, hidden, seed=4)).cuda()
    s, nd = ev.run(params, torch.tensor(feats).cuda(), feats.shape[0], torch.tensor(ids).cuda(), torch.tensor(labels).cuda())
    torch.cuda.synchronize()
    assert np.isfinite(nd.cpu().numpy()).all()
    masked = ev.masked.cpu().numpy()
    assert (masked[4] == -100000.0).all() and (masked[9, 7:] == -100000.0).all()  # remove_padding_for_metric_eval
    np.testing.assert_allclose(s.cpu().numpy(), scores, atol=1e-6)                  # validation returns UNMASKED scores


def test_all_zero_labels_na():
    """NA with no click anywhere: w = 1e-7 on every document (base_algorithm.py:321), D = B*L*1e-7 - finite, uniform targets."""
    rng = np.random.RandomState(2)
    F, hidden, B, L = 16, [24], 6, 8
    feats = rng.uniform(-1, 1, size=(B * L, F)).astype(np.float32)
    ids = rng.permutation(B * L).astype(np.int32).reshape(L, B)
    _step(F, hidden, B, L, feats, ids, np.zeros((L, B), np.float32), None)


@pytest.mark.parametrize("algo,L", [("softmax", 256), ("dla", 256), ("pairdebias", 256), ("lambdarank", 256)])
def test_longest_lists(algo, L):
    """list_size 256 - the cap of the list-wise kernels (one wavefront x 4 documents per lane)."""
    from oracle import ultr_oracle as O
    rng = np.random.RandomState(L)
    B = 3
    scores = rng.normal(size=(B, L)).astype(np.float32)
    labels_LB = (rng.uniform(size=(L, B)) < 0.2).astype(np.float32)
    if algo == "lambdarank":
        labels_LB = rng.randint(0, 5, size=(L, B)).astype(np.float32)
    labels_LB[0, :] = 1.0
    run = HipRun(8, [4], B, L, algo=algo)
    run.set_inputs(np.zeros((1, 8), np.float32), np.zeros((L, B), np.int32), labels_LB)
    s = torch.tensor(scores, requires_grad=True)
    y = torch.tensor(labels_LB.T.copy())
    tp = torch.tensor(rng.uniform(0.8, 1.2, size=L).astype(np.float32))
    tm = torch.tensor(rng.uniform(0.8, 1.2, size=L).astype(np.float32))
    if algo == "softmax":
        ipw = rng.uniform(1, 10, size=40)
        loss = O.softmax_loss(s, y, O.ipw_weights(labels_LB, ipw))
        ds, tail = run.loss(ipw_table=ipw, scores=scores)
        gs = 1.0 / tail[1]
    elif algo == "dla":
        q = torch.tensor(rng.normal(scale=0.3, size=L + 1).astype(np.float32))
        with torch.no_grad():
            pw = O.normalized_weights(torch.softmax(O.denoising_net(q, B, L), -1))
        loss = O.softmax_loss(s, y, pw)
        ds, tail = run.loss(aux=q.numpy(), scores=scores)
        gs = 1.0 / tail[1]
    elif algo == "pairdebias":
        loss = O.pairdebias_loss(s, torch.tensor(labels_LB), tp, tm)[0]
        ds, tail = run.loss(aux=np.concatenate([tp.numpy(), tm.numpy()]), scores=scores)
        gs = 1.0
    else:
        loss = O.lambdarank_loss(s, y, tp, tm, 1.0)[0]
        ds, tail = run.loss(aux=np.concatenate([tp.numpy(), tm.numpy()]), scores=scores)
        gs = 1.0 / tail[1]
    (g,) = torch.autograd.grad(loss, s)
    g = g.numpy()
    np.testing.assert_allclose(ds * gs, g, rtol=3e-5, atol=3e-6 * max(1.0, float(np.abs(g).max())))


def test_weight_copy_staleness_guard(monkeypatch):
    """The forward reads a k-major COPY of the hidden weights.  In-place tensor ops bump torch's version counter and rebuild
    it; a write through `.data` does not - ULTR_CHECK_WT=1 turns that into an error, invalidate_weight_copy() is the fix."""
    from ultra_pytorch_amd import hip_ops
    from ultra_pytorch_amd.ranking_model import DNN
    monkeypatch.setenv("ULTR_CHECK_WT", "1")
    model = DNN("hidden_layer_sizes=[16, 8]", 12).cuda()
    x = [torch.randn(5, 12, device="cuda") for _ in range(3)]
    s0 = torch.cat(model.build(x), dim=1).clone()
    w = model.sequential.linear0.weight
    w.copy_(w * 2.0)                      # in place: version counter moves, copy rebuilt
    s1 = torch.cat(model.build(x), dim=1).clone()
    assert not torch.allclose(s0, s1)
    w.data.mul_(0.5)                      # behind the counter: the copy is stale ...
    with pytest.raises(RuntimeError, match="stale k-major weight copy"):
        model.build(x)
    model.invalidate_weight_copy()        # ... until the writer says so
    s2 = torch.cat(model.build(x), dim=1)
    torch.testing.assert_close(s2, s0, rtol=1e-5, atol=1e-6)


def test_host_mapped_step_report_equals_device_scalars():
    """StepEngine.read_scalars() (the update kernel's report in host-mapped pinned memory, what the plugins' loss read uses
    instead of loss.item()) returns exactly the device-side scalars, for consecutive steps, without a stream sync."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops, synthetic
    F, hidden, B, L = 24, [16, 8], 12, 6
    rng = np.random.RandomState(5)
    shape = hip_ops.DnnShape(F, hidden, "elu")
    eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax")
    p = dev(O.init_params(F, hidden, seed=3))
    st = torch.zeros_like(p)
    for k in range(5):
        feats, ids, y = synthetic.make_batch(rng, B, L, F)
        eng.train_step(p, st, dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y))
        got = eng.read_scalars()
        torch.cuda.synchronize()
        want = eng.scalars[:8].cpu().numpy()
        assert np.array_equal(got, want), (k, got, want)
        assert np.isfinite(got[0]) and eng.read_loss() == float(want[0])


@pytest.mark.parametrize("F,hidden,B,L", [(136, [256, 256], 33, 10),    # config 2's layers, ragged batch (fragment-major path)
                                          (40, [512, 256], 7, 10),      # 512-wide rows: the two-chunk-per-lane build of the fused kernel
                                          (24, [64, 32, 32], 20, 8),    # three hidden layers; 64 / 32-wide: split contraction (k-major path)
                                          (136, [128], 16, 16),         # one hidden layer, full 16-row tiles
                                          (32, [96, 64], 12, 5),        # three lists per 16-row tile
                                          (20, [48, 24], 9, 3)])        # widths not multiples of 32: no fragment-major copies
def test_fused_step_shapes_match_oracle(F, hidden, B, L, mfma_mode, wgrad_path):
    """The small-batch NA / IPW step (ONE fused forward + loss + backward launch, dnn_fb_kernel) through ultr_train_step at
    shapes that reach every path of the kernel - fragment-major and k-major weight streaming, one / two column chunks per
    lane, 1 - 5 lists per tile - against the oracle: scores, loss, gradient, norm, updated parameters."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops
    rng = np.random.RandomState(F + B)
    n_docs = B * L - 2
    feats = rng.uniform(-1, 1, size=(n_docs, F)).astype(np.float32)
    ids = rng.permutation(B * L)
    ids = np.where(ids >= n_docs, n_docs, ids).astype(np.int32).reshape(L, B)  # two PAD documents
    clicks = (rng.uniform(size=(L, B)) < 0.35).astype(np.float32)
    clicks[0, :] = 1.0
    params = O.init_params(F, hidden, seed=7)
    for name, shape, off in O.param_layout(F, hidden):  # non-trivial LayerNorm affine parameters
        if "layer_norm" in name:
            n = int(np.prod(shape))
            params[off:off + n] += rng.uniform(-0.3, 0.3, size=n).astype(np.float32)
    state0 = (0.01 * rng.uniform(size=params.shape)).astype(np.float32)
    ipw = np.linspace(1.0, 6.0, 12)
    ref = O.train_step_softmax(params, state0, F, hidden, feats, ids, clicks, ipw_list=ipw)
    shape = hip_ops.DnnShape(F, hidden, "elu")
    eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax")
    p, st = dev(params), dev(state0)
    eng.train_step(p, st, dev(feats), n_docs, dev(ids, torch.int32), dev(clicks), ipw_table=dev(ipw.astype(np.float32)))
    sc = eng.read_scalars()
    np.testing.assert_allclose(eng.scores.cpu().numpy(), ref["scores"], atol=1e-5, rtol=1e-5)
    assert abs(sc[0] - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
    g = eng.grads[:shape.n_params].cpu().numpy() / sc[3]
    np.testing.assert_allclose(g, ref["grads"], rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(ref["grads"]).max())))
    assert abs(sc[1] - ref["norm"]) <= 1e-5 * max(1.0, ref["norm"])
    sel = np.abs(ref["grads"]) > 1e-4 * np.abs(ref["grads"]).max()
    np.testing.assert_allclose(p.cpu().numpy()[sel], ref["params"][sel], rtol=1e-5, atol=5e-6)
    # the step that follows streams the weight copies the update kernel maintained (k-major, image, fragment-major): its scores
    # must be the oracle's forward on the oracle's updated parameters (they differ from the kernel's by the 1e-5 bar)
    eng.train_step(p, st, dev(feats), n_docs, dev(ids, torch.int32), dev(clicks), ipw_table=dev(ipw.astype(np.float32)))
    sc2 = eng.read_scalars()
    ref2 = O.train_step_softmax(ref["params"], ref["state"], F, hidden, feats, ids, clicks, ipw_list=ipw)
    np.testing.assert_allclose(eng.scores.cpu().numpy(), ref2["scores"], atol=2e-4, rtol=2e-4)
    assert abs(sc2[0] - ref2["loss"]) <= 1e-4 * max(1.0, abs(ref2["loss"]))


def _softmax_case(F, hidden, B, L):
    from oracle import ultr_oracle as O
    rng = np.random.RandomState(F + B)
    n_docs = B * L - 2
    feats = rng.uniform(-1, 1, size=(n_docs, F)).astype(np.float32)
    ids = rng.permutation(B * L)
    ids = np.where(ids >= n_docs, n_docs, ids).astype(np.int32).reshape(L, B)  # two PAD documents
    clicks = (rng.uniform(size=(L, B)) < 0.35).astype(np.float32)
    clicks[0, :] = 1.0
    params = O.init_params(F, hidden, seed=7)
    for name, shape, off in O.param_layout(F, hidden):  # non-trivial LayerNorm affine parameters
        if "layer_norm" in name:
            n = int(np.prod(shape))
            params[off:off + n] += rng.uniform(-0.3, 0.3, size=n).astype(np.float32)
    state0 = (0.01 * rng.uniform(size=params.shape)).astype(np.float32)
    return n_docs, feats, ids, clicks, params, state0


@pytest.mark.parametrize("F,hidden,B,L", [(136, [256, 256], 33, 10),       # config 2's layers: both products of every layer on the split-half copies
                                          (136, [512, 256, 128], 21, 20),  # config 3's layers: split-half where a layer has >= 256 outputs, fp32 elsewhere
                                          (40, [512, 256], 7, 10),         # 512-wide rows
                                          (24, [64, 32, 32], 20, 8)])      # no split-half copies at all
def test_separate_kernel_step_shapes_match_oracle(F, hidden, B, L, mfma_mode, wgrad_path, monkeypatch):
    """The same step through the SEPARATE forward / backward kernels (dnn_fwd_kernel, dnn_bwd2_kernel: what every batch too
    large for the fused kernel runs, forced here with ULTR_NO_FUSED_FB=1) - including their split-half (fp16 hi/lo) products -
    against the oracle: scores, loss, gradient, norm, updated parameters, and the scores of the step after."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops
    monkeypatch.setenv("ULTR_NO_FUSED_FB", "1")
    n_docs, feats, ids, clicks, params, state0 = _softmax_case(F, hidden, B, L)
    ipw = np.linspace(1.0, 6.0, 24)
    ref = O.train_step_softmax(params, state0, F, hidden, feats, ids, clicks, ipw_list=ipw)
    shape = hip_ops.DnnShape(F, hidden, "elu")
    try:
        eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax")  # re-reads the knobs
        p, st = dev(params), dev(state0)
        eng.train_step(p, st, dev(feats), n_docs, dev(ids, torch.int32), dev(clicks), ipw_table=dev(ipw.astype(np.float32)))
        sc = eng.read_scalars()
        np.testing.assert_allclose(eng.scores.cpu().numpy(), ref["scores"], atol=1e-5, rtol=1e-5)
        assert abs(sc[0] - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
        g = eng.grads[:shape.n_params].cpu().numpy() / sc[3]
        np.testing.assert_allclose(g, ref["grads"], rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(ref["grads"]).max())))
        assert abs(sc[1] - ref["norm"]) <= 1e-5 * max(1.0, ref["norm"])
        eng.train_step(p, st, dev(feats), n_docs, dev(ids, torch.int32), dev(clicks), ipw_table=dev(ipw.astype(np.float32)))
        sc2 = eng.read_scalars()
        ref2 = O.train_step_softmax(ref["params"], ref["state"], F, hidden, feats, ids, clicks, ipw_list=ipw)
        np.testing.assert_allclose(eng.scores.cpu().numpy(), ref2["scores"], atol=2e-4, rtol=2e-4)
        assert abs(sc2[0] - ref2["loss"]) <= 1e-4 * max(1.0, abs(ref2["loss"]))
    finally:
        monkeypatch.delenv("ULTR_NO_FUSED_FB")
        shape.lib.ultr_config_reload()


@pytest.mark.parametrize("hidden", [[256, 64], [256, 256], [512, 256, 128]])
def test_update_kernel_keeps_every_weight_copy_current(hidden):
    """k-major copy, packed image, the fragment-major copies and the split-half (fp16 hi/lo) copies after ultr_apply_update ==
    a fresh ultr_dnn_build_wt of the updated parameters (bitwise): a model with fragment-major copies only, config 2's layers
    (split-half copies of every product) and config 3's (split-half copies of some layers); ragged F."""
    import ctypes
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops, synthetic
    F, B, L = 136, 16, 10
    rng = np.random.RandomState(1)
    shape = hip_ops.DnnShape(F, hidden, "elu")
    eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax")
    p = dev(O.init_params(F, hidden, seed=9))
    st = torch.zeros_like(p)
    for _ in range(3):
        feats, ids, y = synthetic.make_batch(rng, B, L, F)
        eng.train_step(p, st, dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y))
    torch.cuda.synchronize()
    kept = hip_ops.weight_copy(shape).get(p).clone()
    fresh = torch.zeros_like(kept)
    from ultra_pytorch_amd._lib import check
    check(shape.lib.ultr_dnn_build_wt(ctypes.byref(shape.desc), ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(fresh.data_ptr()),
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ultr_dnn_build_wt")
    torch.cuda.synchronize()
    assert kept.numel() > shape.n_params  # copies + image + fragment-major region
    diff = (kept != fresh).nonzero().flatten()
    assert diff.numel() == 0, (diff[:8].tolist(), diff.numel(), kept.numel())


@pytest.mark.parametrize("F,hidden,B,L,separate", [(136, [256, 256], 256, 10, False),      # config 2: the fused kernel
                                                   (136, [256, 256], 256, 10, True),       # config 2 through the separate kernels
                                                   (136, [512, 256, 128], 512, 20, True),  # config 3 (640 workgroups of the row-tile kernels)
                                                   (700, [512, 256, 128], 256, 50, True)])  # config 4's grid (800 row tiles)
def test_repeated_launches_are_bitwise_identical(F, hidden, B, L, separate, mfma_mode, monkeypatch):
    """The same step from the same state, a hundred times: scores, loss and every gradient bit must repeat - with the split-half
    products and with every product on the fp32 matrix cores.  (Round 3 met gradients that differed in a few launches out of
    ten on grids of >= ~20 workgroups, invisible to a single parity run.  Round 4 found the cause - a gfx950 hazard of packed
    fp32 instructions with op_sel next to another wave's f16 MFMAs, profiles/r04_h3_rootcause.md - and builds the library
    without packed fp32 instructions; this test and its long version tools/h3_stress.py keep watch.)"""
    from ultra_pytorch_amd import engine, hip_ops, synthetic
    from ultra_pytorch_amd.ranking_model import init_flat_params
    if separate:
        monkeypatch.setenv("ULTR_NO_FUSED_FB", "1")
    shape = hip_ops.DnnShape(F, hidden, "elu")
    try:
        feats, ids, y = synthetic.make_batch(np.random.RandomState(5), B, L, F)
        ipw = np.asarray(synthetic.load_ipw(), np.float32)
        p0 = init_flat_params(shape, seed=3).numpy()
        eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax", learning_rate=0.05, max_gradient_norm=5.0)
        f, i, yy, tab = dev(feats), dev(ids, torch.int32), dev(y), dev(ipw)
        first = None
        for rep in range(100):
            params, state = dev(p0.copy()), dev(np.zeros_like(p0))
            eng.train_step(params, state, f, feats.shape[0], i, yy, ipw_table=tab)
            torch.cuda.synchronize()
            got = (eng.grads[:shape.n_params].clone(), eng.scores.clone(), params.clone())
            if first is None:
                first = got
                continue
            for a, b, what in zip(got, first, ("gradients", "scores", "updated parameters")):
                assert torch.equal(a, b), (rep, what, float((a - b).abs().max()))
    finally:
        if separate:
            monkeypatch.delenv("ULTR_NO_FUSED_FB")
            shape.lib.ultr_config_reload()


def _h3_case(weight, B=16):
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import synthetic
    F, hidden, L = 136, [256, 256], 10
    p0 = O.init_params(F, hidden, seed=3)
    for name, shp, off in O.param_layout(F, hidden):
        if name.endswith("linear1.weight"):
            p0[off + 5] = weight
    feats, ids, y = synthetic.make_batch(np.random.RandomState(2), B, L, F)
    return F, hidden, L, p0, feats, ids, y


@pytest.mark.parametrize("weight", [70.0, 200.0])
@pytest.mark.parametrize("reader", ["read_loss", "read_scalars"])
def test_weight_outside_the_split_half_range_falls_back_to_fp32(weight, reader, monkeypatch):
    """The wide layers' products read fp16 hi / lo copies of the weights x 2^8 (|w| < 128).  The reference trains any weight
    magnitude (base_algorithm.py:208-226): parameters that ARRIVE with a hidden weight >= 64 (near the edge) or >= 128 (beyond it)
    make the engine switch to the fp32 matrix-core products - with a warning - before a kernel reads the copies, and the step
    matches the oracle at the usual bars."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops
    for k in ("ULTR_FB_H3", "ULTR_FWD_H3", "ULTR_BWD_H3"):
        monkeypatch.setenv(k, "1")
    F, hidden, L, p0, feats, ids, y = _h3_case(weight)
    B = ids.shape[1]
    shape = hip_ops.DnnShape(F, hidden, "elu")
    try:
        eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax")
        p, st = dev(p0), dev(np.zeros_like(p0))
        with pytest.warns(RuntimeWarning, match="fp32 matrix cores"):
            eng.train_step(p, st, dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y))
        loss = eng.read_loss() if reader == "read_loss" else float(eng.read_scalars()[0])
        # THIS model is on the fp32 products now; the process-wide knobs and the environment are untouched, and a second model
        # of the same process keeps the split-half plan
        assert not hip_ops.split_half_enabled(shape) and hip_ops.split_half_enabled()
        assert all(os.environ.get(k) == "1" for k in ("ULTR_FB_H3", "ULTR_FWD_H3", "ULTR_BWD_H3"))
        assert hip_ops.split_half_enabled(hip_ops.DnnShape(F, hidden, "elu"))
        ref = O.train_step_softmax(p0, np.zeros_like(p0), F, hidden, feats, ids, y, ipw_list=None)
        assert abs(loss - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
        np.testing.assert_allclose(eng.scores.cpu().numpy(), ref["scores"], atol=1e-5 * max(1.0, float(np.abs(ref["scores"]).max())), rtol=1e-5)
        sc = eng.read_scalars()
        g = eng.grads[:shape.n_params].cpu().numpy() / sc[3]
        np.testing.assert_allclose(g, ref["grads"], rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(ref["grads"]).max())))
    finally:
        monkeypatch.undo()
        shape.lib.ultr_config_reload()


@pytest.mark.parametrize("reader", ["read_loss", "read_scalars"])
def test_weight_drifting_towards_the_split_half_range_switches_plans_in_time(reader, monkeypatch):
    """Training drift: a hidden weight just below 64 is pushed over it by the optimizer.  The update kernel raises
    ULTR_STATUS_H3_NEAR in the step report, the next read of the loss (read_loss: the ADVICE r03 path that never looked at the
    status) switches THIS MODEL to the fp32 products with a warning, nothing raises, and the trajectory goes on finite: every
    copy was still exact when the switch happened (|w| < 128)."""
    from ultra_pytorch_amd import engine, hip_ops
    for k in ("ULTR_FB_H3", "ULTR_FWD_H3", "ULTR_BWD_H3"):
        monkeypatch.setenv(k, "1")
    F, hidden, L, p0, feats, ids, y = _h3_case(0.01)
    B = ids.shape[1]
    shape = hip_ops.DnnShape(F, hidden, "elu")
    try:
        # eight weights of layer 1 start a hair inside +-64 (different inputs k: their gradients' signs are unrelated); plain SGD
        # moves each by lr x g per step, so within a few steps some of them cross
        from oracle import ultr_oracle as O
        for name, shp, off in O.param_layout(F, hidden):
            if name.endswith("linear1.weight"):
                for t in range(8):
                    p0[off + 256 * (3 + t) + 11 * t] = (63.9999 if t % 2 == 0 else -63.9999)
        assert float(np.abs(p0).max()) < 64.0
        eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax", optimizer="sgd", learning_rate=2.0)
        p, st = dev(p0), dev(np.zeros_like(p0))
        args = (dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y))
        import warnings
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            seen = False
            for step in range(6):
                eng.train_step(p, st, *args)
                loss = eng.read_loss() if reader == "read_loss" else float(eng.read_scalars()[0])
                assert np.isfinite(loss)
                seen = seen or any("fp32 matrix cores" in str(x.message) for x in w)
            torch.cuda.synchronize()
        wmax = float(p.abs().max())
        assert wmax >= 64.0, "the planted weights did not cross 64: the test needs a larger learning rate (%.6f)" % wmax
        assert seen and not hip_ops.split_half_enabled(shape) and hip_ops.split_half_enabled()
        assert wmax < 128.0 and np.isfinite(p.cpu().numpy()).all()
    finally:
        monkeypatch.undo()
        shape.lib.ultr_config_reload()


def test_fp32_fallback_of_one_model_leaves_a_second_engine_consistent(monkeypatch):
    """the automatic fp32 fallback is per MODEL (ultr_dnn_desc::flags), not a process-wide knob flip.  Two
    engines in one process: model A arrives with a hidden weight of 200 (falls back), model B is ordinary.  B's step before and
    after A's fallback is bit-identical (same plan, same kernels), B still reports split-half products, and A matches the oracle."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops
    for k in ("ULTR_FB_H3", "ULTR_FWD_H3", "ULTR_BWD_H3"):
        monkeypatch.setenv(k, "1")
    F, hidden, L, pA, feats, ids, y = _h3_case(200.0)
    _, _, _, pB, _, _, _ = _h3_case(0.01)
    B = ids.shape[1]
    shA, shB = hip_ops.DnnShape(F, hidden, "elu"), hip_ops.DnnShape(F, hidden, "elu")
    try:
        engA = engine.StepEngine(shA, B, L, torch.device("cuda"), algo="softmax")
        engB = engine.StepEngine(shB, B, L, torch.device("cuda"), algo="softmax")
        args = (dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y))

        def step_b():
            p, st = dev(pB), dev(np.zeros_like(pB))
            engB.train_step(p, st, *args)
            engB.read_scalars()
            return engB.scores.cpu().numpy().copy(), engB.grads.cpu().numpy().copy(), p.cpu().numpy().copy()

        before = step_b()
        pa, sa = dev(pA), dev(np.zeros_like(pA))
        with pytest.warns(RuntimeWarning, match="fp32 matrix cores"):
            engA.train_step(pa, sa, *args)
        lossA = engA.read_loss()
        assert not hip_ops.split_half_enabled(shA) and hip_ops.split_half_enabled(shB)
        after = step_b()
        for a, b in zip(before, after):
            assert np.array_equal(a, b)
        ref = O.train_step_softmax(pA, np.zeros_like(pA), F, hidden, feats, ids, y, ipw_list=None)
        assert abs(lossA - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
    finally:
        monkeypatch.undo()
        shA.lib.ultr_config_reload()


@pytest.mark.parametrize("hidden", [[128, 64], [256, 100]])
def test_big_path_planes_of_narrow_layers_are_range_checked(hidden, monkeypatch):
    """the per-layer big-batch path builds split-half planes of EVERY hidden layer - also of layers without
    fragment copies (fewer than 256 outputs, or a width that is no multiple of 32) - so the range of every hidden weight is
    tracked: a model of such widths that arrives with a weight of 200 is switched to the fp32 products before a kernel reads a
    plane, and the forced per-layer step matches the oracle (it used to run on overflowed planes: NaN gradients, silently)."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops, synthetic
    for k in ("ULTR_FB_H3", "ULTR_FWD_H3", "ULTR_BWD_H3"):
        monkeypatch.setenv(k, "1")
    monkeypatch.setenv("ULTR_BIG_FWD", "2")
    monkeypatch.setenv("ULTR_BIG_BWD", "2")
    F, L, B = 136, 10, 64
    p0 = O.init_params(F, hidden, seed=4)
    for name, shp, off in O.param_layout(F, hidden):
        if name.endswith("linear1.weight"):
            p0[off + 3] = 200.0
    feats, ids, y = synthetic.make_batch(np.random.RandomState(6), B, L, F)
    shape = hip_ops.DnnShape(F, hidden, "elu")
    try:
        eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax")
        p, st = dev(p0), dev(np.zeros_like(p0))
        with pytest.warns(RuntimeWarning, match="fp32 matrix cores"):
            eng.train_step(p, st, dev(feats), feats.shape[0], dev(ids, torch.int32), dev(y))
        sc = eng.read_scalars()
        assert not hip_ops.split_half_enabled(shape)
        ref = O.train_step_softmax(p0, np.zeros_like(p0), F, hidden, feats, ids, y, ipw_list=None)
        assert abs(float(sc[0]) - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
        g = eng.grads[:shape.n_params].cpu().numpy() / sc[3]
        assert np.isfinite(g).all()
        # (a weight of 200 in front of a LayerNorm: the bar of the full-size tests, 1e-5 relative + 1e-5 of the largest entry)
        np.testing.assert_allclose(g, ref["grads"], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(ref["grads"]).max())))
    finally:
        monkeypatch.undo()
        shape.lib.ultr_config_reload()


@pytest.mark.parametrize("separate", [False, True])
def test_split_half_products_with_wide_dynamic_range(separate, wgrad_path, monkeypatch):
    """The error budget of the split-half (fp16 hi / lo) products at its edges: feature rows with a 1e4
    outlier next to 1e-4 entries (LayerNorm_0 bounds what reaches the product, but the row's scale is set by its largest
    element), all-zero rows, weights from 1e-7 to 60 in one layer, LayerNorm gains up to 30 - scores, loss and gradients against
    the oracle at the standard 1e-5 bars."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops
    if separate:
        monkeypatch.setenv("ULTR_NO_FUSED_FB", "1")
    F, hidden, B, L = 136, [256, 256], 24, 10
    rng = np.random.RandomState(11)
    n_docs = B * L
    feats = rng.uniform(-1, 1, size=(n_docs, F)).astype(np.float32)
    feats[::7, 3] = 1e4                      # outliers
    feats[::5, 10:40] *= 1e-4                # tiny entries in the same rows as others of order 1
    feats[13] = 0.0                          # an all-zero document
    ids = rng.permutation(n_docs).astype(np.int32).reshape(L, B)
    clicks = (rng.uniform(size=(L, B)) < 0.3).astype(np.float32)
    clicks[0, :] = 1.0
    params = O.init_params(F, hidden, seed=5)
    for name, shape, off in O.param_layout(F, hidden):
        n = int(np.prod(shape))
        if name.endswith("linear1.weight"):
            w = params[off:off + n]
            w[::97] = 60.0 * np.sign(w[::97] + 1e-9)
            w[1::89] *= 1e-6
        if name.endswith("layer_norm1.weight"):
            params[off:off + n:11] = 30.0
    state0 = np.zeros_like(params)
    ref = O.train_step_softmax(params, state0, F, hidden, feats, ids, clicks, ipw_list=None)
    shape = hip_ops.DnnShape(F, hidden, "elu")
    try:
        eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax")
        p, st = dev(params), dev(state0)
        eng.train_step(p, st, dev(feats), n_docs, dev(ids, torch.int32), dev(clicks))
        sc = eng.read_scalars()
        smax = max(1.0, float(np.abs(ref["scores"]).max()))
        np.testing.assert_allclose(eng.scores.cpu().numpy(), ref["scores"], atol=1e-5 * smax, rtol=1e-5)
        assert abs(sc[0] - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
        g = eng.grads[:shape.n_params].cpu().numpy() / sc[3]
        gmax = float(np.abs(ref["grads"]).max())
        err = np.abs(g - ref["grads"]).max() / max(gmax, 1e-30)
        margins.check("edges/split_half_dynamic_range" + ("_separate" if separate else "") + ("" if wgrad_path == "slabs" else "_wgrad_" + wgrad_path),
                      "grads_max_abs_diff_over_max_abs_g", err)
        np.testing.assert_allclose(g, ref["grads"], rtol=1e-5, atol=1e-5 * max(1.0, gmax))
    finally:
        if separate:
            monkeypatch.delenv("ULTR_NO_FUSED_FB")
            shape.lib.ultr_config_reload()


@pytest.mark.parametrize("act", ["relu", "tanh", "sigmoid"])
@pytest.mark.parametrize("separate", [False, True])
def test_wide_layers_with_every_activation_match_oracle(act, separate, monkeypatch):
    """The reference-fixture nets for relu / tanh / sigmoid are narrow (no split-half products, no fragment-major copies): the same
    activations on config 2's layers - through the fused kernel and through the separate forward / backward kernels - against
    the oracle: scores, loss, gradients at the standard bars."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import engine, hip_ops
    if separate:
        monkeypatch.setenv("ULTR_NO_FUSED_FB", "1")
    F, hidden, B, L = 136, [256, 256], 40, 10
    n_docs, feats, ids, clicks, params, state0 = _softmax_case(F, hidden, B, L)
    ipw = np.linspace(1.0, 4.0, 10)
    ref = O.train_step_softmax(params, state0, F, hidden, feats, ids, clicks, ipw_list=ipw, act=act)
    shape = hip_ops.DnnShape(F, hidden, act)
    try:
        eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax")
        p, st = dev(params), dev(state0)
        eng.train_step(p, st, dev(feats), n_docs, dev(ids, torch.int32), dev(clicks), ipw_table=dev(ipw.astype(np.float32)))
        sc = eng.read_scalars()
        np.testing.assert_allclose(eng.scores.cpu().numpy(), ref["scores"], atol=1e-5, rtol=1e-5)
        assert abs(sc[0] - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
        g = eng.grads[:shape.n_params].cpu().numpy() / sc[3]
        np.testing.assert_allclose(g, ref["grads"], rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(ref["grads"]).max())))
    finally:
        if separate:
            monkeypatch.delenv("ULTR_NO_FUSED_FB")
            shape.lib.ultr_config_reload()


def test_split_half_weight_gradients_across_a_wide_range_of_dz(monkeypatch):
    """dnn_wgrad_h3_kernel's scale is ONE power of two per 64 columns of an operand and wave group: rows of dz that differ by
    2^20 inside a workgroup's walk (the scale is lowered on the way and the sums follow exactly) and a block of columns 2^16 below its neighbours.
    Every entry of dW_1 against an fp64 product of the same dz and u within 1e-5 x (|entry| + sum of |terms|) - the bar of the full-size tests -
    through the stage API with hand-made dscores-free inputs: forward, then ultr_dnn_backward on a crafted dscores vector."""
    from oracle import ultr_oracle as O
    from tests.hipref import HipRun
    monkeypatch.setenv("ULTR_WG_H3", "2")
    monkeypatch.setenv("ULTR_NO_FUSED_FB", "1")
    F, hidden, B, L = 136, [256, 128], 40, 10
    rng = np.random.RandomState(3)
    n_docs = B * L
    feats = rng.uniform(-1, 1, size=(n_docs, F)).astype(np.float32)
    ids = rng.permutation(n_docs).astype(np.int32).reshape(L, B)
    params = O.init_params(F, hidden, seed=2)
    for name, shape, off in O.param_layout(F, hidden):
        if name.endswith("linear1.weight"):  # output units 0..31 of layer 1 see 2^-16 of the gradient of the others
            w = params[off:off + int(np.prod(shape))].reshape(shape)
            w[:32] *= 2.0 ** -16
    try:
        run = HipRun(F, hidden, B, L, algo="softmax")
        run.set_inputs(feats, ids, np.zeros((L, B), np.float32))
        run.forward(params)
        ds = rng.normal(size=(B, L)).astype(np.float32)
        ds[: B // 2] *= 2.0 ** -20  # the first half of the lists: gradients a million times smaller
        g, _ = run.backward(dscores=ds)
        x = O.gather_rows(feats, ids).numpy()
        ref = O.dnn_backward_manual(params, F, hidden, x, ds.T.reshape(-1))
        terms = O.dnn_backward_manual(params, F, hidden, x, ds.T.reshape(-1), abs_terms=True)
        bad = np.abs(g - ref) > 1e-5 * (np.abs(ref) + terms)
        assert not bad.any(), (int(bad.sum()), float((np.abs(g - ref) / np.maximum(terms, 1e-30)).max()))
    finally:
        monkeypatch.undo()
        from ultra_pytorch_amd import _lib
        _lib.load().ultr_config_reload()
