"""SURVEY Appendix A - the reference quirks that affect parity, one explicit test each (CPU: oracle + host logic).
The GPU-side counterparts live in test_gpu_parity.py / test_gpu_plugins.py (golden vectors exercise all of them)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import ultr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a1_layernorm_before_every_linear():
    """A.1  DNN.py:43-55: one LayerNorm per Linear, the scorer included."""
    names = [n for n, _, _ in O.param_layout(136, [32, 16])]
    assert names == ["sequential.layer_norm0.weight", "sequential.layer_norm0.bias", "sequential.linear0.weight",
                     "sequential.linear0.bias", "sequential.layer_norm1.weight", "sequential.layer_norm1.bias",
                     "sequential.linear1.weight", "sequential.linear1.bias", "sequential.layer_norm2.weight",
                     "sequential.layer_norm2.bias", "sequential.linear2.weight", "sequential.linear2.bias"]


def test_a2_softmax_smoothing_and_global_normaliser():
    """A.2  base_algorithm.py:324-330: labels + 1e-7, and ONE normaliser for the whole batch (not per list)."""
    rng = np.random.RandomState(0)
    s = torch.from_numpy(rng.normal(size=(3, 5)).astype(np.float32))
    y = torch.tensor([[1., 0, 0, 0, 0], [1, 1, 1, 0, 0], [0, 0, 0, 0, 0]])  # third list: only the smoothing term
    loss = float(O.softmax_loss(s, y))
    w = y + 1e-7
    ls = torch.log_softmax(s, dim=1)
    per_list = -(w * ls).sum(1)
    assert abs(loss - float(per_list.sum() / w.sum())) < 1e-6          # global normaliser
    assert abs(loss - float((per_list / w.sum(1)).mean())) > 1e-3      # not the per-list mean
    assert float(per_list[2]) > 0.0                                     # the all-zero list still contributes (1e-7)


def test_a4_ipw_unclicked_zero_and_table_saturation():
    """A.4  propensity_estimator.py:36-41: weight 0 without a click; positions past the table reuse its LAST entry."""
    clicks = np.zeros((6, 2), np.float32)
    clicks[0, 0] = clicks[5, 0] = clicks[3, 1] = 1.0
    pw = O.ipw_weights(clicks, [1.0, 2.0, 4.0]).numpy()
    assert pw.shape == (2, 6)
    assert pw[0, 0] == 1.0 and pw[0, 5] == 4.0 and pw[1, 3] == 4.0     # saturated index
    assert pw.sum() == 9.0                                              # everything else 0


def test_a5_dla_stateless_adagrad_is_sign_like():
    """A.5  dla.py:153-154: optimizers are rebuilt every step, so Adagrad's accumulator is empty when step() runs:
    p -= lr * g / (|g| + 1e-10) - the step size does not depend on |g|."""
    p = torch.zeros(4)
    g = torch.tensor([1e-3, -2.0, 5.0, -1e-4])
    p2, _, _, _ = O.apply_update(p, g, torch.full((4,), 123.0), lr=0.05, max_norm=0.0, stateless=True)
    np.testing.assert_allclose(p2.numpy(), -0.05 * np.sign(g.numpy()), rtol=1e-5)


def test_a5_denoising_net_is_batch_independent():
    """A.5  dla.py:24-48: the propensity logits depend on the position only."""
    pp = torch.arange(6, dtype=torch.float32) * 0.1  # W[5] | b
    out = O.denoising_net(pp, 3, 5)
    assert tuple(out.shape) == (3, 5) and torch.equal(out[0], out[1]) and torch.equal(out[1], out[2])


def test_a6_pairdebias_batch_inflation_and_lr_default():
    """A.6  base_algorithm.py:242-248: the [B] * [B,1] broadcast multiplies the pair loss by B; default lr 0.005."""
    from ultra_pytorch_amd.utils import HParams
    rng = np.random.RandomState(1)
    B, L = 4, 3
    s = torch.from_numpy(rng.normal(size=(B, L)).astype(np.float32))
    c = torch.from_numpy((rng.uniform(size=(L, B)) < 0.5).astype(np.float32))
    c[0, :] = 1.0
    c[1, :] = 0.0
    t = torch.ones(1, L)
    loss, PL, _, _ = O.pairdebias_loss(s, c, t, t)
    # hand count: sum over ordered pairs (i, j), clicked i / unclicked j, of softplus(s_j - s_i), times B
    ref = 0.0
    for b in range(B):
        for i in range(L):
            for j in range(L):
                if i != j and c[i, b] > c[j, b]:
                    ref += float(torch.nn.functional.softplus(s[b, j] - s[b, i]))
    assert abs(float(loss) - B * ref) < 1e-4 * max(1.0, abs(B * ref))
    import ultra_pytorch_amd.learning_algorithm.pairwise_debias as pd
    src = open(pd.__file__).read()
    assert "learning_rate=0.005" in src.replace(" ", "") or "learning_rate=0.005" in src


def test_a7_lambdarank_bce_on_probabilities_and_natural_log_idcg():
    """A.7  lambda_rank.py:128, 247-266: BCE-WITH-LOGITS applied to a probability; the IDCG uses ln (not log2) and is
    ONE scalar for the whole batch."""
    s = torch.tensor([[2.0, 1.0, 0.5], [0.1, 0.3, 0.2]])
    y = torch.tensor([[2.0, 0.0, 1.0], [0.0, 1.0, 0.0]])
    t = torch.ones(1, 3)
    loss, PL, _, _ = O.lambdarank_loss(s, y, t, t, 1.0)
    assert np.isfinite(float(loss)) and float(loss) > 0
    ideal = torch.sort(y, dim=1, descending=True).values
    idcg = float(((2.0 ** ideal - 1) / torch.log(torch.arange(3, dtype=torch.float32) + 2)).sum())  # natural log, batch sum
    # doubling every gain by the same idcg: recompute the loss with labels of ONE list changed must change the other list's terms
    y2 = y.clone()
    y2[1] = torch.tensor([0.0, 3.0, 0.0])
    loss2, PL2, _, _ = O.lambdarank_loss(s, y2, t, t, 1.0)
    assert abs(float(loss2) - float(loss)) > 1e-6 and idcg > 0


def test_a8_l2_loss_is_rejected():
    """A.8  ipw_rank.py:154-159: with l2_loss > 0 the reference silently skips gradient clipping; not supported here."""
    import ultra_pytorch_amd.learning_algorithm.base_algorithm as ba
    assert "l2_loss" in open(ba.__file__).read()


def test_a10_validation_scores_are_not_masked():
    """A.10  base_algorithm.py:88-116: metrics see -100000 at PAD positions, the caller gets the raw scores."""
    rng = np.random.RandomState(2)
    F, hidden = 8, [4]
    p = O.init_params(F, hidden, seed=0)
    feats = rng.uniform(-1, 1, size=(5, F)).astype(np.float32)
    ids = np.array([[0, 2], [1, 3], [5, 4], [5, 5]], np.int32)  # 5 == n_docs == PAD
    lab = np.array([[1, 0], [0, 2], [0, 1], [0, 0]], np.float32)
    r = O.validation(p, F, hidden, feats, ids, lab, topn=(1, 3))
    assert (r["masked"][ids.T == 5] == O.PADDING_SCORE).all()
    assert (r["scores"][ids.T == 5] != O.PADDING_SCORE).all() and np.isfinite(r["scores"]).all()


def test_a11_ndcg_topn_clipped_and_invalid_labels():
    """A.11  metrics.py:224-265: topn is clipped to the list size; labels < 0 count as 0 and sink to rowmin - 1e-6."""
    lab = torch.tensor([[2.0, -1.0, 1.0]])
    sc = torch.tensor([[0.1, 9.0, 0.5]])
    n = O.ndcg(lab, sc, [1, 10])
    l2, p2, topn = O._prepare(lab, sc, [1, 10])
    assert topn == [1, 3] and float(l2[0, 1]) == 0.0 and float(p2[0, 1]) < float(p2.min()) + 1e-5
    assert 0.0 < float(n[0]) <= 1.0 and 0.0 < float(n[1]) <= 1.0


def test_a12_stop_condition_only_at_checkpoints():
    """A.12  main.py:162,221: `current_step > max_train_iteration` is only tested every steps_per_checkpoint steps."""
    import ultra_pytorch_amd.main as m
    src = open(m.__file__).read()
    i_chk = src.index("if current_step % args.steps_per_checkpoint != 0:")
    i_stop = src.index("if args.max_train_iteration > 0 and current_step > args.max_train_iteration:")
    assert i_chk < i_stop


def test_a13_direct_label_feed_may_return_fewer_lists():
    """A.13  direct_label_feed.py:117-126: with check_validation an all-zero list is skipped and NOT replaced."""
    from ultra_pytorch_amd.input_layer import DirectLabelFeed
    from ultra_pytorch_amd.utils.data_utils import Raw_data

    class Algo:
        rank_list_size, max_candidate_num, feature_size = 3, 3, 2
        letor_features_name = "letor_features"
        docid_inputs_name = ["docid_input%d" % i for i in range(3)]
        labels_name = ["label%d" % i for i in range(3)]

    ds = Raw_data()
    ds.feature_size, ds.rank_list_size = 2, 3
    ds.features = [[0.1, 0.2]] * 6
    ds.dids = ["d%d" % i for i in range(6)]
    ds.qids = ["q0", "q1"]
    ds.initial_list = [[0, 1, 2], [3, 4, 5]]
    ds.labels = [[1, 0, 0], [0, 0, 0]]  # the second list has no relevant document
    ds.initial_list_lengths = [3, 3]
    ds.pad(3)
    feed = DirectLabelFeed(Algo(), 8, "")
    random.seed(0)
    input_feed, info = feed.get_batch(ds, check_validation=True)
    n_lists = len(np.asarray(input_feed["docid_input0"]))
    assert 0 < n_lists < 8 and len(info["rank_list_idxs"]) == 8


def test_a14_ranklist_scores_are_plain_numbers(tmp_path):
    """A.14  data_utils.py:638-639: the reference prints `tensor(x)`; here a TREC-readable float."""
    from ultra_pytorch_amd.utils import data_utils as du
    ds = du.Raw_data()
    ds.qids, ds.dids = ["q0"], ["d0", "d1"]
    ds.initial_list = [[0, 1]]
    du.output_ranklist(ds, [[0.25, 0.75]], str(tmp_path) + "/", "t")
    lines = open(str(tmp_path) + "/t.ranklist").read().strip().split("\n")
    assert len(lines) == 2 and "tensor" not in lines[0]
    assert lines[0].split()[2] == "d1" and float(lines[0].split()[4]) == 0.75


def test_reference_structure_variants_equal_the_vectorised_oracle():
    """bench.py times the oracle in the reference's own structure as well: the 2-level Python pair loop of
    PairDebias and DLA's per-step torch.optim.Adagrad objects.  Both must be the same computation."""
    rng = np.random.RandomState(0)
    F_, hidden, B, L = 9, [7, 5], 5, 4
    feats = rng.uniform(-1, 1, size=(B * L, F_)).astype(np.float32)
    ids = np.arange(B * L, dtype=np.int64).reshape(B, L).T.copy()
    clicks = (rng.uniform(size=(L, B)) < 0.5).astype(np.float32)
    clicks[0, :] = np.arange(B) % 2
    clicks[1, :] = 1 - clicks[0, :]
    p0 = O.init_params(F_, hidden, seed=2)
    s0 = (0.01 * rng.uniform(size=p0.shape)).astype(np.float32)
    tp, tm = np.linspace(0.9, 1.1, L).astype(np.float32), np.linspace(1.1, 0.9, L).astype(np.float32)
    a = O.pairdebias_step(p0, s0, tp, tm, F_, hidden, feats, ids, clicks)
    b = O.pairdebias_step(p0, s0, tp, tm, F_, hidden, feats, ids, clicks, loops=True)
    assert abs(a["loss"] - b["loss"]) <= 1e-5 * abs(a["loss"])
    np.testing.assert_allclose(a["grads"], b["grads"], rtol=1e-4, atol=1e-5 * np.abs(a["grads"]).max())
    np.testing.assert_allclose(a["t_plus"], b["t_plus"], atol=1e-6)
    np.testing.assert_allclose(a["t_minus"], b["t_minus"], atol=1e-6)
    q0 = (0.1 * rng.randn(L + 1)).astype(np.float32)
    c = O.dla_step(p0, q0, F_, hidden, feats, ids, clicks)
    d = O.dla_step(p0, q0, F_, hidden, feats, ids, clicks, fresh_optimizers=True)
    sel = np.abs(c["grads"]) > 1e-6 * np.abs(c["grads"]).max()  # the sign-like update is ill-conditioned at g ~ 0
    np.testing.assert_allclose(c["params"][sel], d["params"][sel], atol=2e-6)
    np.testing.assert_allclose(c["prop_params"], d["prop_params"], atol=2e-6)
    assert abs(c["norm"] - d["norm"]) < 1e-5 and abs(c["prop_norm"] - d["prop_norm"]) < 1e-6


@pytest.mark.parametrize("algo", ["softmax", "dla", "pairdebias", "lambdarank"])
def test_torch_baseline_matches_oracle(algo):
    """tools/torch_rocm_baseline.py (bench.py's `torch_rocm_baseline` leg: the reference's step in stock torch ops, timed on the
    GPU for context) computes the oracle's step: same loss and same post-step parameters on CPU."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch_rocm_baseline as TB
    from ultra_pytorch_amd import synthetic
    Fs, L, B, hidden = 12, 6, 5, [8, 4]
    rng = np.random.RandomState(3)
    feats, ids, y = synthetic.make_batch(rng, B, L, Fs, clicks=(algo != "lambdarank"))
    if algo == "pairdebias":  # position 0 must take part in pairs on both sides (EM ratios are normalised by its sums)
        y[0, :] = np.arange(B) % 2
        y[1, :] = 1.0 - y[0, :]
    params = O.init_params(Fs, hidden, seed=2)
    ipw = [1.0, 2.0, 3.5]
    cfg = dict(F=Fs, L=L, B=B, hidden=hidden, algo=algo, model="dnn")
    lr = 0.005 if algo == "pairdebias" else 0.05
    st = TB.Stepper(cfg, params, ipw if algo == "softmax" else None, torch.device("cpu"), lr)
    loss = st.step(st.stage((feats, ids, y)))
    got = torch.cat([p.detach().reshape(-1) for p in st.model.parameters()]).numpy()
    z = np.zeros_like(params)
    if algo == "softmax":
        r = O.train_step_softmax(params, z, Fs, hidden, feats, ids, y, ipw_list=ipw)
    elif algo == "dla":
        r = O.dla_step(params, np.zeros(L + 1, np.float32), Fs, hidden, feats, ids, y)
    elif algo == "pairdebias":
        r = O.pairdebias_step(params, z, np.ones(L, np.float32), np.ones(L, np.float32), Fs, hidden, feats, ids, y)
    else:
        r = O.lambdarank_step(params, z, np.ones(L, np.float32), np.ones(L, np.float32), Fs, hidden, feats, ids, y)
    assert abs(loss - r["loss"]) <= 1e-5 * max(1.0, abs(r["loss"]))
    sel = np.abs(r["grads"]) > 1e-4 * np.abs(r["grads"]).max()  # first Adagrad step is sign-like where g ~ 0
    np.testing.assert_allclose(got[sel], r["params"][sel], rtol=1e-4, atol=1e-5)
    if algo in ("pairdebias", "lambdarank"):
        np.testing.assert_allclose(st.tp.numpy(), r["t_plus"].ravel(), atol=1e-5)
