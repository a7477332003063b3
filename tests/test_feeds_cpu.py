"""Host-side callers (data loader, click model, feeds, ranklist writer) against what the REFERENCE's own loader /
feeds produced on the same toy dataset and Python random stream (tests/golden/feeds_toy.npz).  CPU only."""
import json
import os
import random

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DATA = os.path.join(GOLDEN, "ultra_toy_data") + "/"


class ModelStub:
    """what a feed reads from the learning-algorithm object (click_simulation_feed.py:64-67,153-156)"""

    def __init__(self, feature_size, rank_list_size, max_candidate_num):
        self.feature_size, self.rank_list_size, self.max_candidate_num = feature_size, rank_list_size, max_candidate_num
        self.letor_features_name = "letor_features"
        self.docid_inputs_name = ["docid_input%d" % i for i in range(max_candidate_num)]
        self.labels_name = ["label%d" % i for i in range(max_candidate_num)]


def arrays(model, feed, L):
    feats = np.asarray(feed["letor_features"], dtype=np.float32)
    ids = np.stack([feed[model.docid_inputs_name[l]] for l in range(L)]).astype(np.int32)
    lab = np.stack([feed[model.labels_name[l]] for l in range(L)]).astype(np.float32)
    return feats, ids, lab


@pytest.fixture(scope="module")
def gold():
    d = np.load(os.path.join(GOLDEN, "feeds_toy.npz"))
    return d, json.loads(str(d["meta"]))


def test_loader_matches_reference(gold):
    from ultra_pytorch_amd import utils
    d, m = gold
    for prefix in ("train", "valid"):
        ds = utils.read_data(DATA, prefix)
        assert ds.feature_size == m["F"]
        assert list(d[prefix + "_qids"]) == ds.qids
        np.testing.assert_array_equal(d[prefix + "_lens"], ds.initial_list_lengths)
        assert int(d[prefix + "_rank_list_size"]) == ds.rank_list_size
        assert int(d[prefix + "_n_features_rows"]) == len(ds.features)
        assert abs(float(d[prefix + "_feature_sum"]) - float(np.asarray(ds.features).sum())) < 1e-9
        for q, (lst, lab) in enumerate(zip(ds.initial_list, ds.labels)):
            np.testing.assert_array_equal(d[prefix + "_lists"][q][: len(lst)], lst)
            np.testing.assert_array_equal(d[prefix + "_labels"][q][: len(lab)], lab)


def test_feeds_match_reference_bit_exact(gold):
    from ultra_pytorch_amd import utils
    from ultra_pytorch_amd.input_layer import ClickSimulationFeed, DirectLabelFeed
    d, m = gold
    train, valid = utils.read_data(DATA, "train"), utils.read_data(DATA, "valid")
    Lmax, L = m["max_candidate_num"], m["L"]
    train.pad(Lmax)
    valid.pad(Lmax)
    model = ModelStub(m["F"], L, Lmax)
    random.seed(m["seed"])
    feed = ClickSimulationFeed(model, 6, "")
    for t in range(2):
        f, info = feed.get_batch(train, check_validation=True)
        fe, ids, lab = arrays(model, f, L)
        np.testing.assert_array_equal(fe, d["click%d_features" % t])
        np.testing.assert_array_equal(ids, d["click%d_docids" % t])
        np.testing.assert_array_equal(lab, d["click%d_labels" % t])
        np.testing.assert_array_equal(info["rank_list_idxs"], d["click%d_idxs" % t])
        assert f["docid_input0"].dtype == np.float32 and f["label0"].dtype == np.float32
    dfeed = DirectLabelFeed(model, 4, "")
    f, info = dfeed.get_next_batch(0, valid, check_validation=False)
    fe, ids, lab = arrays(model, f, Lmax)
    np.testing.assert_array_equal(fe, d["direct_features"])
    np.testing.assert_array_equal(ids, d["direct_docids"])
    np.testing.assert_array_equal(lab, d["direct_labels"])
    assert (ids == fe.shape[0]).any()  # the toy lists are ragged: pad id == n_docs is present
    random.seed(m["seed"] + 1)
    f, info = dfeed.get_batch(train, check_validation=True)
    fe, ids, lab = arrays(model, f, Lmax)
    np.testing.assert_array_equal(ids, d["directrand_docids"])
    np.testing.assert_array_equal(lab, d["directrand_labels"])


def test_short_list_raises(gold):
    from ultra_pytorch_amd import utils
    from ultra_pytorch_amd.input_layer import ClickSimulationFeed
    d, m = gold
    train = utils.read_data(DATA, "train")  # NOT padded
    model = ModelStub(m["F"], m["L"] + 50, m["max_candidate_num"] + 50)
    with pytest.raises(ValueError):
        ClickSimulationFeed(model, 2, "").get_batch(train, check_validation=True)


def test_click_models_and_estimator():
    from ultra_pytorch_amd import synthetic
    from ultra_pytorch_amd.utils import click_models as cm
    from ultra_pytorch_amd.utils.propensity_estimator import RandomizedPropensityEstimator
    pbm = cm.loadModelFromJson(json.load(open(synthetic.PBM_JSON)))
    random.seed(3)
    clicks, exam, cp = pbm.sampleClicksForOneList([4, 0, 2, 1, 0, 0, 0, 0, 0, 0, 3, 3])
    assert len(clicks) == 12 and exam[11] == exam[9] == 0.06 and cp[0] == 1.0 and cp[1] == 0.1
    est = RandomizedPropensityEstimator(synthetic.IPW_JSON)
    w = est.getPropensityForOneList([1, 0, 1] + [0] * 40 + [1])
    assert w[0] == est.IPW_list[0] and w[1] == 0.0 and w[2] == est.IPW_list[2] and w[-1] == est.IPW_list[-1]


def test_click_models_draw_the_reference_stream():
    """PBM, the user-browsing model and the cascade model against the reference's own simulators (tests/golden/click_models.npz:
    click_models.py:68-110, 113-186, 187-236 run on its example JSONs): same `random` seed -> the same clicks, examination and
    click probabilities at every position, the same propensity weights, and the stream stands at the same place afterwards."""
    from ultra_pytorch_amd import synthetic
    from ultra_pytorch_amd.utils import click_models as cm
    d = np.load(os.path.join(GOLDEN, "click_models.npz"))
    m = json.loads(str(d["meta"]))
    data = os.path.dirname(synthetic.PBM_JSON)
    for key in m["models"]:
        model = cm.loadModelFromJson(json.load(open(os.path.join(data, m["files"][key]))))
        assert model.model_name == {"pbm": "position_biased_model", "ubm": "user_browsing_model", "cascade": "cascade_model"}[key]
        random.seed(m["seed"])
        for i in range(m["n_lists"]):
            c, e, p = model.sampleClicksForOneList(d["labels%d" % i].tolist())
            np.testing.assert_array_equal(np.asarray(c, np.float64), d["%s_clicks%d" % (key, i)])
            np.testing.assert_array_equal(np.asarray(e, np.float64), d["%s_exam%d" % (key, i)])
            np.testing.assert_array_equal(np.asarray(p, np.float64), d["%s_cprob%d" % (key, i)])
            for flag in (False, True):
                np.testing.assert_array_equal(np.asarray(model.estimatePropensityWeightsForOneList([int(x) for x in c], flag), np.float64),
                                              d["%s_pw%d_%d" % (key, i, int(flag))])
        assert random.random() == float(d["%s_next_uniform" % key])


def test_dataset_roundtrip_merge_and_ranklist(tmp_path):
    from ultra_pytorch_amd import utils
    from ultra_pytorch_amd.utils import data_utils
    rng = np.random.RandomState(0)
    feats = np.round(rng.uniform(-1, 1, size=(9, 5)), 6)
    lists, labels = [[0, 1, 2], [3, 4, 5, 6], [7], [8, 2]], [[1, 0, 2], [0, 0, 0, 0], [1], [0, 3]]
    root = str(tmp_path) + "/"
    json.dump({"feature_size": 5, "max_label": 3.0}, open(root + "settings.json", "w"))
    data_utils.write_ultra_dataset(root, "train", feats, lists, labels)
    ds = utils.read_data(root, "train")
    assert ds.qids == ["0", "3"]  # query 1 has no relevant doc, query 2 has < 2 docs (data_utils.py:390)
    assert ds.rank_list_size == 4  # the maximum is taken before invalid queries are removed, as in the reference
    np.testing.assert_allclose(np.asarray(ds.features), feats, atol=1e-6)
    ds.pad(5)
    assert ds.initial_list[0] == [0, 1, 2, -1, -1] and ds.features[-1] == [0.0] * 5
    merged = utils.merge_Summary([{"ndcg_1": 1.0}, {"ndcg_1": 0.0}], [3, 1])
    assert abs(merged["ndcg_1"] - 0.75) < 1e-9
    os.makedirs(root + "out", exist_ok=True)
    utils.output_ranklist(ds, [[0.1, 0.9, 0.5, 7.0, 7.0], [0.3, 0.2, 0, 0, 0]], root + "out/", "train")
    lines = open(root + "out/train.ranklist").read().strip().split("\n")
    assert lines[0].split()[:4] == ["0", "Q0", "train_1", "1"] and len(lines) == 5
