"""Device-side click simulation (SURVEY 8f.2 "next" row): exact structure (ids / pads / rejection), determinism, and
DISTRIBUTIONAL parity with the host ClickSimulationFeed (which is bit-exact with the reference's feed)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class DS:
    def __init__(self, n_queries, lens, F, seed):
        rng = np.random.RandomState(seed)
        self.feature_size, self.features, self.initial_list, self.labels, self.dids, self.qids = F, [], [], [], [], []
        d = 0
        for q in range(n_queries):
            n = int(rng.randint(lens[0], lens[1] + 1))
            self.features += rng.uniform(-1, 1, size=(n, F)).astype(np.float32).tolist()
            self.initial_list.append(list(range(d, d + n)))
            lab = rng.randint(0, 5, size=n)
            lab[0] = max(lab[0], 1)
            self.labels.append([int(v) for v in lab])
            self.dids += ["d%d" % i for i in range(d, d + n)]
            self.qids.append("q%d" % q)
            d += n
        self.rank_list_size = max(len(x) for x in self.initial_list)

    def pad(self, L):
        self.features.append([0.0] * self.feature_size)
        self.initial_list = [x + [-1] * (L - len(x)) for x in self.initial_list]


class Model:
    def __init__(self, F, L):
        self.feature_size, self.rank_list_size, self.max_candidate_num = F, L, L
        self.cuda = torch.device("cuda")
        self.letor_features_name = "letor_features"
        self.docid_inputs_name = ["docid_input%d" % i for i in range(L)]
        self.labels_name = ["label%d" % i for i in range(L)]


def test_structure_determinism_and_distribution():
    from ultra_pytorch_amd.input_layer import ClickSimulationFeed, DeviceClickFeed
    F, L, B = 8, 12, 4096
    ds = DS(300, (4, 12), F, seed=1)
    ds.pad(L)
    model = Model(F, L)
    feed = DeviceClickFeed(model, B, "", seed=7)
    f0, info0 = feed.get_batch(ds, check_validation=True)
    ids0, ck0, q0 = f0["docids"].cpu().numpy().copy(), f0["labels"].cpu().numpy().copy(), info0["rank_list_idxs"].cpu().numpy().copy()
    n_docs = f0["n_docs"]
    assert n_docs == len(ds.dids)
    lists = np.asarray(ds.initial_list)
    expect = np.where(lists[q0] < 0, n_docs, lists[q0]).T  # [L, B]
    np.testing.assert_array_equal(ids0, expect)                 # ids are exactly the sampled queries' lists, PAD = n_docs
    assert (ck0.sum(0) > 0).all()                               # click-less lists were redrawn
    # (pads CAN be clicked: the reference feed samples every position with label 0 for pads)
    assert set(np.unique(ck0)) <= {0.0, 1.0}
    f1, _ = feed.get_batch(ds, check_validation=True)
    assert not np.array_equal(f1["docids"].cpu().numpy(), ids0)  # next step -> different batch
    feed2 = DeviceClickFeed(model, B, "", seed=7)
    g0, _ = feed2.get_batch(ds, check_validation=True)
    assert np.array_equal(g0["docids"].cpu().numpy(), ids0) and np.array_equal(g0["labels"].cpu().numpy(), ck0)  # f(seed, step)
    # uniform query pick
    counts = np.bincount(q0, minlength=300)
    assert counts.min() > 0 and abs(counts.mean() - B / 300) < 1e-9 and counts.max() < 5 * B / 300
    # distributional parity with the host feed (= the reference's feed): per-position click rates
    dev_clicks = [ck0]
    for _ in range(7):
        f, _ = feed.get_batch(ds, check_validation=True)
        dev_clicks.append(f["labels"].cpu().numpy().copy())
    dev_rate = np.concatenate(dev_clicks, axis=1).mean(1)
    random.seed(5)
    host = ClickSimulationFeed(model, 4096, "")
    host_clicks = []
    for _ in range(8):
        f, _ = host.get_batch(ds, check_validation=True)
        host_clicks.append(np.stack([f[model.labels_name[l]] for l in range(L)]))
    host_rate = np.concatenate(host_clicks, axis=1).mean(1)
    n = 8 * 4096
    sigma = np.sqrt(np.maximum(host_rate * (1 - host_rate), 1e-4) * 2 / n)
    assert np.all(np.abs(dev_rate - host_rate) < 5 * sigma + 1e-3), (dev_rate, host_rate)


def test_training_with_resident_dataset():
    """IPWrank trained from the device feed: features uploaded once, one click kernel + one train-step call per step."""
    import json
    from ultra_pytorch_amd.input_layer import DeviceClickFeed
    from ultra_pytorch_amd.utils import find_class
    F, L, B = 16, 10, 64
    ds = DS(200, (10, 10), F, seed=2)
    ds.pad(L)
    exp = {"learning_algorithm": "ultra_pytorch_amd.learning_algorithm.IPWrank", "learning_algorithm_hparams": "",
           "ranking_model": "ultra_pytorch_amd.ranking_model.DNN", "ranking_model_hparams": "hidden_layer_sizes=[32,16]",
           "max_candidate_num": L, "selection_bias_cutoff": L, "metrics": ["ndcg"], "metrics_topn": [1, 3]}
    algo = find_class(exp["learning_algorithm"])(ds, exp)
    feed = DeviceClickFeed(algo, B, "", seed=1)
    losses = []
    for _ in range(60):
        input_feed, _ = feed.get_batch(ds, check_validation=True)
        loss, _, _ = algo.train(input_feed)
        losses.append(loss)
    assert all(np.isfinite(losses)) and np.mean(losses[-10:]) < np.mean(losses[:10])


def _rates(feed, ds, model, L, n_batches, host):
    out = []
    for _ in range(n_batches):
        f, _ = feed.get_batch(ds, check_validation=True)
        out.append(np.stack([f[model.labels_name[l]] for l in range(L)]) if host else f["labels"].cpu().numpy().copy())
    return np.concatenate(out, axis=1)


def test_cascade_model_on_the_device():
    """The cascade click model (click_models.py:187-236) through ultr_click_batch: at most ONE click per list, and it sits where
    the host feed (bit-exact with the reference's) puts it - the distribution of the clicked position agrees to 5 sigma."""
    import os
    from ultra_pytorch_amd import synthetic
    from ultra_pytorch_amd.input_layer import ClickSimulationFeed, DeviceClickFeed
    F, L, B = 8, 12, 4096
    ds = DS(300, (4, 12), F, seed=3)
    ds.pad(L)
    model = Model(F, L)
    hp = "click_model_json=%s" % os.path.join(os.path.dirname(synthetic.PBM_JSON), "cascade_0.1_1.0_4_1.0.json")
    dev_clicks = _rates(DeviceClickFeed(model, B, hp, seed=11), ds, model, L, 8, host=False)
    assert set(np.unique(dev_clicks)) <= {0.0, 1.0}
    assert (dev_clicks.sum(0) == 1.0).all()  # exactly one click: lists without one are redrawn, everything behind the first is dropped
    random.seed(9)
    host_clicks = _rates(ClickSimulationFeed(model, B, hp), ds, model, L, 8, host=True)
    assert (host_clicks.sum(0) == 1.0).all()
    n = dev_clicks.shape[1]
    dr, hr = dev_clicks.mean(1), host_clicks.mean(1)
    sigma = np.sqrt(np.maximum(hr * (1 - hr), 1e-4) * 2 / n)
    assert np.all(np.abs(dr - hr) < 5 * sigma + 1e-3), (dr, hr)


def test_dynamic_bias_eta_change_on_the_device():
    """dynamic_bias_eta_change / dynamic_bias_step_interval (click_simulation_feed.py:165-172): every `interval` batches eta moves
    and the examination table becomes ORIGINAL ** eta.  The device feed follows the host feed: same eta after the same number
    of batches, and the per-position click rates of the batches drawn under the NEW eta agree to 5 sigma (and differ from
    the rates under the old eta, so the change is visible)."""
    from ultra_pytorch_amd.input_layer import ClickSimulationFeed, DeviceClickFeed
    F, L, B = 8, 10, 4096
    ds = DS(300, (10, 10), F, seed=4)
    ds.pad(L)
    model = Model(F, L)
    hp = "dynamic_bias_eta_change=1.5,dynamic_bias_step_interval=4"
    dfeed = DeviceClickFeed(model, B, hp, seed=5)
    random.seed(6)
    hfeed = ClickSimulationFeed(model, B, hp)
    d_old, h_old = _rates(dfeed, ds, model, L, 4, host=False), _rates(hfeed, ds, model, L, 4, host=True)
    assert abs(dfeed.click_model.eta - 2.5) < 1e-12 and abs(hfeed.click_model.eta - 2.5) < 1e-12
    np.testing.assert_allclose(dfeed.exam.cpu().numpy(), np.asarray(hfeed.click_model.exam_prob, np.float32), rtol=1e-6)
    d_new, h_new = _rates(dfeed, ds, model, L, 3, host=False), _rates(hfeed, ds, model, L, 3, host=True)
    for dv, hv in ((d_old, h_old), (d_new, h_new)):
        dr, hr = dv.mean(1), hv.mean(1)
        sigma = np.sqrt(np.maximum(hr * (1 - hr), 1e-4) * 2 / dv.shape[1])
        assert np.all(np.abs(dr - hr) < 5 * sigma + 1e-3), (dr, hr)
    assert np.abs(d_new.mean(1) - d_old.mean(1)).max() > 0.02  # a steeper position bias


@pytest.mark.parametrize("table", ["reference_json", "six_rows_steep", "three_rows_eta2"])
def test_user_browsing_model_on_the_device(table, tmp_path):
    """The user-browsing model (click_models.py:113-186: examination depends on the rank AND the distance to the last click)
    through ultr_click_batch: per-position click rates and the distribution of the GAP between consecutive clicks (what the
    rank x distance table shapes) agree with the host feed - bit-exact with the reference's - to 5 sigma; lists of 14 documents
    reach beyond the table (its last row is reused).  Three tables: the reference's json, a 6-row table with a steep distance
    decay and other click probabilities, a 3-row table with eta = 2."""
    import json
    import os
    from ultra_pytorch_amd import synthetic
    from ultra_pytorch_amd.input_layer import ClickSimulationFeed, DeviceClickFeed
    F, L, B = 8, 14, 4096
    ds = DS(300, (8, 14), F, seed=8)
    ds.pad(L)
    model = Model(F, L)
    path = os.path.join(os.path.dirname(synthetic.PBM_JSON), "ubm_0.1_1_4_1.0.json")
    if table != "reference_json":
        rows, eta, clickp = (6, 1.0, [0.05, 0.2, 0.45, 0.7, 0.95]) if table == "six_rows_steep" else (3, 2.0, [0.1, 0.3, 0.5, 0.7, 0.9])
        rt = np.random.RandomState(rows)
        exam = [[round(float(0.98 ** r * (0.55 ** (r - d) if table == "six_rows_steep" else 0.8 ** (r - d)) * rt.uniform(0.9, 1.0)), 4)
                 for d in range(r + 1)] for r in range(rows)]  # row r: distance to the last click r - d ... 0 (the reference's triangle)
        path = str(tmp_path / "ubm.json")
        with open(path, "w") as f:
            json.dump({"click_prob": clickp, "eta": eta, "exam_prob": exam, "model_name": "user_browsing_model"}, f)
    hp = "click_model_json=%s" % path
    dev_clicks = _rates(DeviceClickFeed(model, B, hp, seed=21), ds, model, L, 8, host=False)
    random.seed(13)
    host_clicks = _rates(ClickSimulationFeed(model, B, hp), ds, model, L, 8, host=True)
    assert set(np.unique(dev_clicks)) <= {0.0, 1.0} and (dev_clicks.sum(0) > 0).all()
    n = dev_clicks.shape[1]

    def close(a, b, what):
        sigma = np.sqrt(np.maximum(b * (1 - b), 1e-4) * 2 / n)
        assert np.all(np.abs(a - b) < 5 * sigma + 1e-3), (what, a, b)

    close(dev_clicks.mean(1), host_clicks.mean(1), "per-position click rates")

    def gap_hist(c):  # P(gap = g) over lists, gap = distance between the first two clicks (0: fewer than two)
        out = np.zeros(L)
        for col in c.T:
            pos = np.flatnonzero(col)
            out[pos[1] - pos[0] if len(pos) >= 2 else 0] += 1
        return out / c.shape[1]

    close(gap_hist(dev_clicks), gap_hist(host_clicks), "gap between the first two clicks")


def test_batches_drawn_ahead_are_the_batches_drawn_on_the_spot():
    """Round 6: a plugin algorithm's train(feed) draws the feed's NEXT batch behind its own step (ultr_feed_train_step,
    DeviceClickFeed.next_click_args); a batch is a pure function of (seed, batch counter), so the sequence of batches must be the
    one a feed hands out when nobody draws ahead - including across a change of get_batch's arguments and a second dataset."""
    from ultra_pytorch_amd.input_layer import DeviceClickFeed
    from ultra_pytorch_amd.utils import find_class
    F, L, B = 16, 10, 64
    ds, ds2 = DS(200, (10, 10), F, seed=3), DS(150, (10, 10), F, seed=4)
    ds.pad(L)
    ds2.pad(L)
    exp = {"learning_algorithm": "ultra_pytorch_amd.learning_algorithm.IPWrank", "learning_algorithm_hparams": "",
           "ranking_model": "ultra_pytorch_amd.ranking_model.DNN", "ranking_model_hparams": "hidden_layer_sizes=[32,16]",
           "max_candidate_num": L, "selection_bias_cutoff": L, "metrics": ["ndcg"], "metrics_topn": [1, 3, 5, 10]}
    algo = find_class(exp["learning_algorithm"])(ds, exp)
    plan = [(ds, True), (ds, True), (ds, True), (ds, False), (ds2, True), (ds2, True), (ds, True)]

    def sequence(train):
        feed = DeviceClickFeed(algo, B, "", seed=11)
        out = []
        for data, check in plan:
            f, info = feed.get_batch(data, check_validation=check)
            out.append((f["docids"].cpu().numpy().copy(), f["labels"].cpu().numpy().copy(), info["rank_list_idxs"].cpu().numpy().copy()))
            if train:
                loss, _, _ = algo.train(f)
                assert np.isfinite(loss)
                # the step consumed THIS batch: its tensors are untouched by the draw of the next one (the other buffer)
                np.testing.assert_array_equal(f["docids"].cpu().numpy(), out[-1][0])
                np.testing.assert_array_equal(f["labels"].cpu().numpy(), out[-1][1])
        return out

    ahead, spot = sequence(True), sequence(False)
    for k, (a, b) in enumerate(zip(ahead, spot)):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y, err_msg="batch %d" % k)


def test_ndcg_report_in_host_mapped_memory_equals_the_two_launch_form():
    """ultr_ndcg_report (one launch, batch means by the last wave, values + sequence word in host-mapped memory) against ultr_ndcg
    (per-list launch + mean launch): identical bits, permutation and masked scores included; repeated launches reuse the counter."""
    from ultra_pytorch_amd import engine, hip_ops, synthetic
    from ultra_pytorch_amd.ranking_model import init_flat_params
    for B, L, F in ((256, 10, 136), (37, 100, 24), (5, 3, 8)):
        shape = hip_ops.DnnShape(F, [32, 16], "elu")
        rng = np.random.RandomState(B)
        p = init_flat_params(shape, 1).cuda()
        ev = engine.EvalEngine(shape, B, L, torch.device("cuda"))
        for rep in range(3):
            feats, ids, y = synthetic.make_batch(rng, B, L, F, clicks=False, n_pad=min(3, L - 1))
            f, i_, y_ = torch.tensor(feats).cuda(), torch.tensor(ids).cuda(), torch.tensor(y).cuda()
            ev.run(p, f, feats.shape[0], i_, y_)
            got = ev.read_ndcg()
            dev_vals = ev.ndcg.cpu().numpy()
            order, masked = ev.order.cpu().numpy().copy(), ev.masked.cpu().numpy().copy()
            ref = torch.zeros(4, device="cuda")
            ws = torch.zeros(B * 4, device="cuda")
            o2 = torch.zeros(B, L, dtype=torch.int32, device="cuda")
            m2 = torch.zeros(B, L, device="cuda")
            hip_ops.ndcg(ev.scores, y_, i_, feats.shape[0], B, L, [1, 3, 5, 10], ref, ws, order_out=o2, masked_out=m2)
            np.testing.assert_array_equal(got, ref.cpu().numpy())
            np.testing.assert_array_equal(dev_vals, got)
            np.testing.assert_array_equal(order, o2.cpu().numpy())
            np.testing.assert_array_equal(masked, m2.cpu().numpy())


@pytest.mark.parametrize("B,L,F,hidden", [(256, 10, 136, [512, 256, 128]), (37, 5, 24, [32, 16]), (9, 1, 8, [16]), (33, 17, 24, [32, 16])])
def test_validation_as_one_host_call_equals_forward_plus_metric(B, L, F, hidden):
    """ultr_dnn_forward_ndcg (EvalEngine.run) against ultr_dnn_forward + ultr_ndcg: scores, masked scores, permutation, per-batch NDCG and
    the host report to the bit - with PAD documents and invalid labels."""
    from ultra_pytorch_amd import engine, hip_ops, synthetic
    from ultra_pytorch_amd.ranking_model import init_flat_params
    shape = hip_ops.DnnShape(F, hidden, "elu")
    p = init_flat_params(shape, 3).cuda()
    dev = torch.device("cuda")
    ev = engine.EvalEngine(shape, B, L, dev)
    r2 = np.random.RandomState(7)
    for rep in range(3):
        feats, ids, y = synthetic.make_batch(r2, B, L, F, clicks=False, n_pad=min(3, L - 1))
        y = y.copy()
        y[r2.rand(*y.shape) < 0.1] = -1.0  # invalid labels (metrics.py:251-264)
        f, i_, y_ = torch.tensor(feats).cuda(), torch.tensor(ids).cuda(), torch.tensor(y).cuda()
        ev.run(p, f, feats.shape[0], i_, y_)
        got = ev.read_ndcg()
        sc = torch.empty(B, L, device=dev)
        hip_ops.dnn_forward(shape, p, f, feats.shape[0], i_, B, L, sc, None)
        assert torch.equal(sc, ev.scores)
        ref, ws = torch.zeros(4, device=dev), torch.zeros(B * 4, device=dev)
        o2, m2 = torch.zeros(B, L, dtype=torch.int32, device=dev), torch.zeros(B, L, device=dev)
        hip_ops.ndcg(sc, y_, i_, feats.shape[0], B, L, [1, 3, 5, 10], ref, ws, order_out=o2, masked_out=m2)
        np.testing.assert_array_equal(got, ref.cpu().numpy())
        np.testing.assert_array_equal(ev.ndcg.cpu().numpy(), got)
        assert torch.equal(ev.order, o2) and torch.equal(ev.masked, m2)
        assert np.isfinite(got).all() and got[-1] > 0
