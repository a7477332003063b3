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
