#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING the reference (ULTRA_pytorch) here.

Runs only in the build container, where /root/reference exists.  It imports the
reference's own `ultra` package (never copied into this repo), drives
`model.train(input_feed)` / `model.validation(input_feed)` on seeded synthetic
data, and records — per step, teacher-forced — the inputs, the pre-state, and
every output the hot path produces (scores, loss, pre-clip grads, grad norm,
post-step params / Adagrad state / EM state, NDCG...).  The resulting `.npz`
files are DATA (inputs + expected outputs) and are committed under
tests/golden/; they are what pins the oracle (oracle/) and the HIP path.

Harness-side shims (reference untouched), per SURVEY.md Appendix B:
  (1) stub `tensorflow` (dead import, ultra/ranking_model/base_ranking_model.py:8)
  (2) stub `torch.utils.tensorboard.SummaryWriter` (tensorboard pkg absent)
  (3) `torch.as_tensor` of list-of-ndarray -> np.asarray first (base_algorithm.py:186)
  (4) `nn.utils.clip_grad_value_` no-op on grad-less tensors (ipw_rank.py:164)

Usage:  python tests/golden/make_golden.py [--only NAME]
"""
import argparse
import importlib.machinery
import io
import contextlib
import json
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def install_shims():
    tf = types.ModuleType("tensorflow")
    tf.__spec__ = importlib.machinery.ModuleSpec("tensorflow", None)
    sys.modules["tensorflow"] = tf
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.__spec__ = importlib.machinery.ModuleSpec("torch.utils.tensorboard", None)

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalars(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def close(self):
            pass

    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    torch.utils.tensorboard = tb
    _as = torch.as_tensor

    def as_tensor(data, dtype=None, device=None):
        if isinstance(data, (list, tuple)) and data and isinstance(data[0], np.ndarray):
            data = np.asarray(data)
        return _as(data, dtype=dtype, device=device)

    torch.as_tensor = as_tensor
    _cgv = nn.utils.clip_grad_value_

    def cgv(params, clip_value, foreach=None):
        params = [params] if isinstance(params, torch.Tensor) else list(params)
        params = [p for p in params if p.grad is not None]
        return None if not params else _cgv(params, clip_value, foreach=foreach)

    nn.utils.clip_grad_value_ = cgv


def import_reference():
    os.chdir(REF)
    sys.path.insert(0, REF)
    install_shims()
    import ultra  # noqa: F401
    import ultra.utils  # noqa: F401
    import ultra.learning_algorithm  # noqa: F401
    import ultra.ranking_model  # noqa: F401
    import ultra.input_layer  # noqa: F401
    return ultra


# ----------------------------------------------------------------------------
# synthetic in-memory dataset (Raw_data() with no args)
# ----------------------------------------------------------------------------
def make_dataset(ultra, seed, n_queries, list_lens, F, max_label=4):
    """list_lens: int (fixed) or (lo, hi) inclusive range of docs per query."""
    rng = np.random.RandomState(seed)
    ds = ultra.utils.data_utils.Raw_data()
    ds.feature_size = F
    feats, init_list, labels, qids, lens = [], [], [], [], []
    did = 0
    for q in range(n_queries):
        n = list_lens if isinstance(list_lens, int) else int(rng.randint(list_lens[0], list_lens[1] + 1))
        f = rng.uniform(-1.0, 1.0, size=(n, F)).astype(np.float32)
        lab = rng.randint(0, max_label + 1, size=n)
        if lab.sum() == 0:
            lab[0] = 1
        feats.extend([[float(v) for v in row] for row in f])
        init_list.append(list(range(did, did + n)))
        labels.append([int(v) for v in lab])
        qids.append("q%d" % q)
        lens.append(n)
        did += n
    ds.features = feats
    ds.dids = ["d%d" % i for i in range(did)]
    ds.qids = qids
    ds.initial_list = init_list
    ds.labels = labels
    ds.initial_list_lengths = lens
    ds.rank_list_size = max(lens)
    ultra.utils.metrics.RankingMetricKey.MAX_LABEL = float(max_label)
    return ds


def flat_state(module):
    return {k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def flat_params(module):
    return np.concatenate([p.detach().cpu().numpy().ravel() for p in module.parameters()]).astype(np.float32)


def adagrad_state(opt, module):
    out = []
    for p in module.parameters():
        st = opt.state.get(p, {})
        if "sum" in st:
            out.append(st["sum"].detach().cpu().numpy().ravel())
        else:
            out.append(np.zeros(p.numel(), np.float32))
    return np.concatenate(out).astype(np.float32)


class Recorder:
    """Hooks into the algorithm object to record scores and pre-clip grads."""

    def __init__(self, algo):
        self.algo = algo
        self.scores = None
        self.clips = []  # list of (flat grads pre-clip, total_norm)
        orig_rm = algo.ranking_model

        def rm(model, list_size):
            out = orig_rm(model, list_size)
            self.scores = out.detach().cpu().numpy().copy()
            return out

        algo.ranking_model = rm
        self._orig_clip = torch.nn.utils.clip_grad_norm_

        def clip(parameters, max_norm, *a, **k):
            ps = [parameters] if isinstance(parameters, torch.Tensor) else list(parameters)
            # l2_loss > 0: the caller's generator is already exhausted by its L2 loop (ipw_rank.py:154-159), the clip sees
            # NO parameters (and returns 0); the gradients are then read off the model
            src = ps if ps else list(self.algo.model.parameters())
            g = np.concatenate([p.grad.detach().cpu().numpy().ravel() for p in src]).astype(np.float32)
            tn = self._orig_clip(ps, max_norm, *a, **k)
            self.clips.append((g, float(tn)))
            return tn

        torch.nn.utils.clip_grad_norm_ = clip
        nn.utils.clip_grad_norm_ = clip

    def reset(self):
        self.scores = None
        self.clips = []

    def close(self):
        torch.nn.utils.clip_grad_norm_ = self._orig_clip
        nn.utils.clip_grad_norm_ = self._orig_clip


def feed_arrays(algo, input_feed, L):
    feats = np.asarray(input_feed["letor_features"], dtype=np.float32)
    if feats.ndim != 2:  # empty batch guard
        feats = feats.reshape(0, algo.feature_size)
    docids = np.stack([np.asarray(input_feed[algo.docid_inputs_name[l]]) for l in range(L)]).astype(np.int32)
    labels = np.stack([np.asarray(input_feed[algo.labels_name[l]]) for l in range(L)]).astype(np.float32)
    return feats, docids, labels


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        return fn(*a, **k)


ALGOS = {
    "na": "ultra.learning_algorithm.NavieAlgorithm",
    "ipw": "ultra.learning_algorithm.IPWrank",
    "dla": "ultra.learning_algorithm.DLA",
    "pairdebias": "ultra.learning_algorithm.PairDebias",
    "lambdarank": "ultra.learning_algorithm.LambdaRank",
    "regem": "ultra.learning_algorithm.RegressionEM",
}


def run_train_case(ultra, name, algo_key, F, L, B, hidden, n_steps, seed, n_queries=64,
                   model_cls="ultra.ranking_model.DNN", algo_hparams="", model_extra=""):
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    ds = make_dataset(ultra, seed, n_queries, L, F)
    exp = {
        "learning_algorithm": ALGOS[algo_key],
        "learning_algorithm_hparams": algo_hparams,
        "ranking_model": model_cls,
        "ranking_model_hparams": ("hidden_layer_sizes=%s" % json.dumps(hidden) if hidden is not None else "") + model_extra,
        "max_candidate_num": L,
        "selection_bias_cutoff": L,
        "metrics": ["ndcg", "mrr", "err"],
        "metrics_topn": [1, 3, 5, 10],
    }
    ds.pad(L)
    algo = quiet(ultra.utils.find_class(exp["learning_algorithm"]), ds, exp)
    if algo_key == "na":
        feed = quiet(ultra.utils.find_class("ultra.input_layer.DirectLabelFeed"), algo, B, "")
    else:
        feed = quiet(ultra.utils.find_class("ultra.input_layer.ClickSimulationFeed"), algo, B, "")
    rec = Recorder(algo)
    out = {"meta": json.dumps({
        "name": name, "algo": algo_key, "F": F, "L": L, "B": B, "hidden": hidden, "n_steps": n_steps,
        "seed": seed, "model": model_cls.rsplit(".", 1)[1], "algo_hparams": algo_hparams,
        "param_keys": list(algo.model.state_dict().keys()),
        "param_shapes": [list(v.shape) for v in algo.model.state_dict().values()],
        "lr": float(algo.learning_rate), "max_gradient_norm": float(algo.hparams.max_gradient_norm),
    })}
    if algo_key == "ipw":
        out["ipw_list"] = np.asarray(algo.propensity_estimator.IPW_list, dtype=np.float64)
    for t in range(n_steps):
        rec.reset()
        # DirectLabelFeed.get_batch may return < B lists (all-zero lists skipped); our
        # synthetic lists always have a positive label so it returns exactly B.
        input_feed, _ = feed.get_batch(ds, check_validation=True)
        feats, docids, labels = feed_arrays(algo, input_feed, L)
        pre = {"params": flat_params(algo.model)}
        if algo_key in ("na", "ipw", "pairdebias", "lambdarank", "regem"):
            pre["adagrad"] = adagrad_state(algo.optimizer_func, algo.model)
        if algo_key == "regem":
            pre["propensity"] = algo.propensity.detach().cpu().numpy().copy()
        if algo_key in ("pairdebias", "lambdarank"):
            pre["t_plus"] = algo.t_plus.detach().cpu().numpy().copy()
            pre["t_minus"] = algo.t_minus.detach().cpu().numpy().copy()
        if algo_key == "dla":
            pre["prop_params"] = flat_params(algo.propensity_model)
        drawn = []
        if algo_key == "regem":
            # RegressionEM draws its Bernoulli pseudo-labels from an unseeded torch.rand (regression_EM.py:30-33):
            # record the uniforms so that the restatement / the kernel can be teacher-forced with the same draw
            orig_rand = torch.rand

            def rec_rand(*a, **k):
                u = orig_rand(*a, **k)
                drawn.append(u.detach().cpu().numpy().copy())
                return u

            torch.rand = rec_rand
        try:
            loss, _, _ = quiet(algo.train, input_feed)
        finally:
            if algo_key == "regem":
                torch.rand = orig_rand
        p = "s%d_" % t
        if algo_key == "regem":
            assert len(drawn) == 1
            out[p + "uniforms"] = drawn[0].astype(np.float32)  # [B, L]
            out[p + "ranker_labels"] = algo.ranker_labels.detach().cpu().numpy().astype(np.float32)
            out[p + "post_propensity"] = algo.propensity.detach().cpu().numpy().copy()
        out[p + "features"] = feats
        out[p + "docids"] = docids
        out[p + "labels"] = labels
        for k, v in pre.items():
            out[p + "pre_" + k] = v
        out[p + "scores"] = rec.scores.astype(np.float32)
        out[p + "loss"] = np.float32(loss)
        out[p + "post_params"] = flat_params(algo.model)
        if algo_key == "dla":
            # clip order in dla.py:161-163: propensity model first, then ranker
            (gp, np_), (gm, nm) = rec.clips
            out[p + "prop_grads"] = gp
            out[p + "prop_norm"] = np.float32(np_)
            out[p + "grads"] = gm
            out[p + "norm"] = np.float32(nm)
            out[p + "post_prop_params"] = flat_params(algo.propensity_model)
            out[p + "rank_loss"] = np.float32(algo.rank_loss.item())
            out[p + "exam_loss"] = np.float32(algo.exam_loss.item())
            out[p + "propensity_weights"] = algo.propensity_weights.detach().cpu().numpy().astype(np.float32)
            out[p + "relevance_weights"] = algo.relevance_weights.detach().cpu().numpy().astype(np.float32)
        else:
            (g, n_), = rec.clips
            out[p + "grads"] = g
            out[p + "norm"] = np.float32(n_)
            out[p + "post_adagrad"] = adagrad_state(algo.optimizer_func, algo.model)
        if algo_key in ("pairdebias", "lambdarank"):
            out[p + "post_t_plus"] = algo.t_plus.detach().cpu().numpy().copy()
            out[p + "post_t_minus"] = algo.t_minus.detach().cpu().numpy().copy()
        if algo_key == "ipw":
            out[p + "pw"] = np.asarray(algo.propensity_weights, dtype=np.float32)  # [B, L]
    rec.close()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(out.items())[:0]})


def run_valid_case(ultra, name, F, Lmax, B, hidden, seed, list_lens, n_queries=24):
    """validation() on ragged lists (pads present): scores, masked scores, metrics, argsort."""
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    ds = make_dataset(ultra, seed, n_queries, list_lens, F)
    exp = {
        "learning_algorithm": ALGOS["ipw"],
        "learning_algorithm_hparams": "",
        "ranking_model": "ultra.ranking_model.DNN",
        "ranking_model_hparams": "hidden_layer_sizes=%s" % json.dumps(hidden),
        "max_candidate_num": Lmax,
        "selection_bias_cutoff": min(10, Lmax),
        "metrics": ["ndcg", "mrr", "err"],
        "metrics_topn": [1, 3, 5, 10],
    }
    ds.pad(Lmax)
    algo = quiet(ultra.utils.find_class(exp["learning_algorithm"]), ds, exp)
    feed = quiet(ultra.utils.find_class("ultra.input_layer.DirectLabelFeed"), algo, B, "")
    out = {"meta": json.dumps({
        "name": name, "F": F, "L": Lmax, "B": B, "hidden": hidden, "seed": seed, "max_label": 4.0,
        "metrics": exp["metrics"], "topn": exp["metrics_topn"],
        "param_keys": list(algo.model.state_dict().keys()),
        "param_shapes": [list(v.shape) for v in algo.model.state_dict().values()],
    })}
    out["params"] = flat_params(algo.model)
    it, bi = 0, 0
    while it < len(ds.initial_list):
        input_feed, info = feed.get_next_batch(it, ds, check_validation=False)
        nb = len(info["input_list"])
        feats, docids, labels = feed_arrays(algo, input_feed, Lmax)
        _, scores, summary = quiet(algo.validation, input_feed)
        p = "b%d_" % bi
        out[p + "features"] = feats
        out[p + "docids"] = docids
        out[p + "labels"] = labels
        out[p + "scores"] = scores.detach().cpu().numpy().astype(np.float32)
        masked = algo.remove_padding_for_metric_eval(algo.docid_inputs, algo.output)
        out[p + "masked_scores"] = masked.detach().cpu().numpy().astype(np.float32)
        # the permutation the metric code sorts with (metrics.py:208), after label validation
        lab_t, pred_t, _, _ = ultra.utils.metrics._prepare_and_validate_params(algo.labels, masked, None, [1])
        out[p + "argsort_desc"] = pred_t.sort(descending=True, dim=-1)[1].numpy().astype(np.int32)
        for k, v in summary.items():
            out[p + "metric_" + k] = np.float32(v)
        it += nb
        bi += 1
    out["n_batches"] = np.int32(bi)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def run_feed_case(ultra, name, seed=0):
    """input_feed dicts the reference's feeds build from the toy ULTRA dataset (tests/data of the reference, copied
    as DATA to tests/golden/ultra_toy_data) with random.seed(seed), plus what its loader parsed."""
    data_dir = os.path.join(HERE, "ultra_toy_data") + "/"
    out = {}
    sets = {}
    for prefix in ("train", "valid"):
        ds = quiet(ultra.utils.read_data, data_dir, prefix, None, None)
        sets[prefix] = ds
        out[prefix + "_qids"] = np.asarray(ds.qids)
        out[prefix + "_lens"] = np.asarray(ds.initial_list_lengths, dtype=np.int32)
        out[prefix + "_rank_list_size"] = np.int32(ds.rank_list_size)
        out[prefix + "_n_features_rows"] = np.int32(len(ds.features))
        out[prefix + "_feature_sum"] = np.float64(np.asarray(ds.features, dtype=np.float64).sum())
        out[prefix + "_lists"] = np.asarray([x + [-7] * (16 - len(x)) for x in ds.initial_list], dtype=np.int32)
        out[prefix + "_labels"] = np.asarray([x + [-7.0] * (16 - len(x)) for x in ds.labels], dtype=np.float32)
    max_cand = max(sets["train"].rank_list_size, sets["valid"].rank_list_size)
    exp = {"learning_algorithm": ALGOS["ipw"], "learning_algorithm_hparams": "", "ranking_model": "ultra.ranking_model.DNN",
           "ranking_model_hparams": "hidden_layer_sizes=[8]", "max_candidate_num": max_cand,
           "selection_bias_cutoff": min(10, max_cand), "metrics": ["ndcg"], "metrics_topn": [1, 3]}
    for ds in sets.values():
        ds.pad(max_cand)
    algo = quiet(ultra.utils.find_class(exp["learning_algorithm"]), sets["train"], exp)
    L = exp["selection_bias_cutoff"]
    random.seed(seed)
    feed = quiet(ultra.utils.find_class("ultra.input_layer.ClickSimulationFeed"), algo, 6, "")
    for t in range(2):
        f, info = feed.get_batch(sets["train"], check_validation=True)
        fe, ids, lab = feed_arrays(algo, f, L)
        out["click%d_features" % t], out["click%d_docids" % t], out["click%d_labels" % t] = fe, ids, lab
        out["click%d_idxs" % t] = np.asarray(info["rank_list_idxs"], dtype=np.int32)
    dfeed = quiet(ultra.utils.find_class("ultra.input_layer.DirectLabelFeed"), algo, 4, "")
    f, info = dfeed.get_next_batch(0, sets["valid"], check_validation=False)
    fe, ids, lab = feed_arrays(algo, f, max_cand)
    out["direct_features"], out["direct_docids"], out["direct_labels"] = fe, ids, lab
    random.seed(seed + 1)
    f, info = dfeed.get_batch(sets["train"], check_validation=True)
    fe, ids, lab = feed_arrays(algo, f, max_cand)
    out["directrand_features"], out["directrand_docids"], out["directrand_labels"] = fe, ids, lab
    out["meta"] = json.dumps({"name": name, "seed": seed, "max_candidate_num": int(max_cand), "L": int(L), "F": int(sets["train"].feature_size)})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def run_driver_case(ultra, name, seed=0):
    """The reference DRIVER (main.py:85-227) end to end on the toy ULTRA dataset (tests/golden/ultra_toy_data = the
    reference's tests/data, stored as data): NA + DNN[32,16] + DirectLabelFeed, batch 8, a checkpoint every 3 steps,
    --max_train_iteration 4 (the stop test only runs at checkpoint boundaries: 6 steps, 2 checkpoints).  Records the
    initial weights (so that the counterpart can be teacher-forced), every step's loss, what was printed at each
    checkpoint (global step, averaged loss, merged validation metrics) and every state_dict handed to torch.save."""
    import runpy
    import tempfile
    data_dir = os.path.join(HERE, "ultra_toy_data") + "/"
    tmp = tempfile.mkdtemp(prefix="ultr_driver_")
    settings = {
        "train_input_feed": "ultra.input_layer.DirectLabelFeed", "train_input_hparams": "",
        "valid_input_feed": "ultra.input_layer.DirectLabelFeed", "valid_input_hparams": "",
        "test_input_feed": "ultra.input_layer.DirectLabelFeed", "test_input_hparams": "",
        "ranking_model": "ultra.ranking_model.DNN", "ranking_model_hparams": "hidden_layer_sizes=[32, 16]",
        "learning_algorithm": "ultra.learning_algorithm.NavieAlgorithm", "learning_algorithm_hparams": "",
        "metrics": ["mrr", "ndcg"], "metrics_topn": [1, 3, 5, 10], "objective_metric": "ndcg_10",
    }
    sf = os.path.join(tmp, "settings.json")
    json.dump(settings, open(sf, "w"))
    argv = ["--data_dir", data_dir, "--setting_file", sf, "--model_dir", tmp + "/model/", "--output_dir", tmp + "/out/",
            "--batch_size", "8", "--max_train_iteration", "4", "--steps_per_checkpoint", "3"]
    os.makedirs(tmp + "/model/", exist_ok=True)
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    cls = ultra.learning_algorithm.NavieAlgorithm
    rec = {"init": None, "losses": [], "saves": []}
    orig_train, orig_save = cls.train, torch.save

    def train(self, input_feed):
        if rec["init"] is None:
            rec["init"] = flat_state(self.model)
        out = orig_train(self, input_feed)
        rec["losses"].append(float(out[0]))
        return out

    def save(obj, path, *a, **k):
        rec["saves"].append((len(rec["losses"]), {kk: vv.detach().cpu().numpy().copy() for kk, vv in obj.items()}))
        return orig_save(obj, path, *a, **k)

    cls.train, torch.save = train, save
    buf = io.StringIO()
    old_argv = sys.argv
    try:
        sys.argv = ["main.py"] + argv
        with contextlib.redirect_stdout(buf):
            runpy.run_path(os.path.join(REF, "main.py"), run_name="__main__")
    finally:
        sys.argv = old_argv
        cls.train, torch.save = orig_train, orig_save
    history = parse_driver_stdout(buf.getvalue())  # what the driver printed at each checkpoint
    out = {"meta": json.dumps({"name": name, "seed": seed, "argv": argv[6:], "settings": settings, "history": history,
                               "save_steps": [s for s, _ in rec["saves"]], "n_steps": len(rec["losses"]),
                               "param_keys": list(rec["init"].keys())})}
    out["losses"] = np.asarray(rec["losses"], np.float64)
    for k, v in rec["init"].items():
        out["init_" + k] = v
    for i, (_, sd) in enumerate(rec["saves"]):
        for k, v in sd.items():
            out["save%d_%s" % (i, k)] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "steps", len(rec["losses"]), "checkpoints at", [s for s, _ in rec["saves"]], history)


def parse_driver_stdout(text):
    """What main.py printed at each checkpoint: (global step, averaged loss, merged validation metrics)."""
    history, cur = [], None
    for line in text.splitlines():
        if line.startswith("global step "):
            tok = line.split()
            cur = {"global_step": int(tok[2]), "loss": float(tok[-1]), "metrics": {}}
            history.append(cur)
        elif cur is not None:
            tok = line.split()
            if len(tok) == 2 and tok[0].rsplit("_", 1)[-1].isdigit() and ":" not in tok[0]:
                try:
                    cur["metrics"][tok[0]] = float(tok[1])
                except ValueError:
                    pass
    return history


CONV_ALGOS = {"ipw": "IPWrank", "dla": "DLA", "pairdebias": "PairDebias"}
CONV_SEEDS = (0, 1, 2, 3, 4)
CONV_TOPN = (1, 3, 5, 10)


def feed_checksum(algo, input_feed, L):
    """Two numbers per batch that pin WHICH documents sit where and WHICH of them were clicked (position-weighted sums)."""
    w = np.arange(1, L + 1, dtype=np.float64)[:, None]
    ids = np.stack([np.asarray(input_feed[algo.docid_inputs_name[l]], np.float64) for l in range(L)])
    lab = np.stack([np.asarray(input_feed[algo.labels_name[l]], np.float64) for l in range(L)])
    col = np.arange(1, ids.shape[1] + 1, dtype=np.float64)[None, :]
    return float((ids * w * col).sum()), float((lab * w * col).sum())


def run_convergence_case(ultra, name, algo_key, n_iter=300, ckpt_every=50, batch=64, n_ulp=1, synth=False, seeds=None):
    """End-of-training NDCG@10 of the reference's own main.py (main.py:85-227) on the toy ULTRA dataset for a click-feed
    algorithm (ipw_rank.py:102-182, dla.py:179-266, pairwise_debias.py:106-174): ClickSimulationFeed (PBM), DNN[32,16],
    batch 64, a checkpoint every 50 steps, --max_train_iteration 300 (the stop test runs at checkpoint boundaries: 350 steps,
    7 checkpoints), 5 seeds.  Per seed THREE runs of the reference: 1 thread, 8 threads, and 1 thread with the initial weights
    moved by one fp32 rounding (x (1 +- 2^-23)): the spread between them is the reference's own sensitivity to summation
    order / rounding, the band a counterpart with different fp32 summation order is held to.  Stores arrays only: initial
    weights (and DLA's propensity parameters), per-step batch checksums and losses of the 1-thread run, validation
    ndcg_{1,3,5,10} at every checkpoint of every variant."""
    import runpy
    import tempfile
    data_dir = os.path.join(HERE, "ultra_toy_data") + "/"
    dataset_meta = {"dir": "ultra_toy_data"}
    if synth:
        # the seeded synthetic dataset (tests/golden/make_synth_dataset.py: 400 validation queries, learnable labels) instead of the
        # reference's 8-query toy set: an end-of-training figure that can tell a wrong trainer from a right one
        sys.path.insert(0, HERE)
        import make_synth_dataset as MS
        data_dir = tempfile.mkdtemp(prefix="ultr_synth_") + "/"
        dataset_meta = {"generator": "tests/golden/make_synth_dataset.py", "seed": MS.SEED, "sha256": MS.write_dataset(data_dir)}
    seeds = tuple(CONV_SEEDS if seeds is None else seeds)
    cls_name = CONV_ALGOS[algo_key]
    settings = {
        "train_input_feed": "ultra.input_layer.ClickSimulationFeed", "train_input_hparams": "",
        "valid_input_feed": "ultra.input_layer.DirectLabelFeed", "valid_input_hparams": "",
        "test_input_feed": "ultra.input_layer.DirectLabelFeed", "test_input_hparams": "",
        "ranking_model": "ultra.ranking_model.DNN", "ranking_model_hparams": "hidden_layer_sizes=[32, 16]",
        "learning_algorithm": "ultra.learning_algorithm." + cls_name, "learning_algorithm_hparams": "",
        "metrics": ["ndcg"], "metrics_topn": list(CONV_TOPN), "objective_metric": "ndcg_10",
    }
    cls = getattr(ultra.learning_algorithm, cls_name)
    out, meta_runs = {}, {}
    # (DLA: n_ulp = 6 - its stateless sign-like updates turn one rounding into a different trajectory, three variants understate the band)
    variants = (("t1", 1, None), ("t8", 8, None)) + tuple(("ulp" if k == 0 else "ulp%d" % (k + 1), 1, k + 1) for k in range(n_ulp))
    for seed in seeds:
        init_sd, init_prop = None, None
        for vname, threads, perturb in variants:
            tmp = tempfile.mkdtemp(prefix="ultr_conv_")
            sf = os.path.join(tmp, "settings.json")
            json.dump(settings, open(sf, "w"))
            os.makedirs(tmp + "/model/", exist_ok=True)
            argv = ["--data_dir", data_dir, "--setting_file", sf, "--model_dir", tmp + "/model/", "--output_dir", tmp + "/out/",
                    "--batch_size", str(batch), "--max_train_iteration", str(n_iter), "--steps_per_checkpoint", str(ckpt_every)]
            torch.set_num_threads(threads)
            torch.manual_seed(seed)
            random.seed(seed)
            np.random.seed(seed)
            rec = {"first": True, "losses": [], "sums": []}
            orig_train = cls.train

            def train(self, input_feed, rec=rec, perturb=perturb, vname=vname):
                nonlocal init_sd, init_prop
                if rec["first"]:
                    rec["first"] = False
                    if init_sd is None:  # first variant of the seed: what torch.manual_seed(seed) initialised
                        init_sd = flat_state(self.model)
                        if hasattr(self, "propensity_model"):
                            init_prop = flat_state(self.propensity_model)
                    else:                # later variants start from the SAME weights (whatever the thread count did to the init)
                        self.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in init_sd.items()})
                        if init_prop is not None:
                            self.propensity_model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in init_prop.items()})
                    if perturb:
                        g = torch.Generator().manual_seed(1000 * perturb + seed)
                        with torch.no_grad():
                            for p in self.model.parameters():
                                sign = (torch.randint(0, 2, p.shape, generator=g).to(p.dtype) * 2 - 1)
                                p.mul_(1.0 + sign * 2.0 ** -23)
                L = self.exp_settings["selection_bias_cutoff"]
                rec["sums"].append(feed_checksum(self, input_feed, L))
                o = orig_train(self, input_feed)
                rec["losses"].append(float(o[0]))
                return o

            cls.train = train
            buf = io.StringIO()
            old_argv = sys.argv
            try:
                sys.argv = ["main.py"] + argv
                with contextlib.redirect_stdout(buf):
                    runpy.run_path(os.path.join(REF, "main.py"), run_name="__main__")
            finally:
                sys.argv = old_argv
                cls.train = orig_train
                torch.set_num_threads(1)
            hist = parse_driver_stdout(buf.getvalue())
            nd = np.asarray([[h["metrics"]["ndcg_%d" % n] for n in CONV_TOPN] for h in hist], np.float64)
            out["s%d_%s_ndcg" % (seed, vname)] = nd
            out["s%d_%s_ckpt_loss" % (seed, vname)] = np.asarray([h["loss"] for h in hist], np.float64)
            out["s%d_%s_losses" % (seed, vname)] = np.asarray(rec["losses"], np.float64)
            if vname == "t1":
                out["s%d_losses" % seed] = np.asarray(rec["losses"], np.float64)
                out["s%d_feed_sums" % seed] = np.asarray(rec["sums"], np.float64)
                for k, v in init_sd.items():
                    out["s%d_init_%s" % (seed, k)] = v
                if init_prop is not None:
                    for k, v in init_prop.items():
                        out["s%d_prop_%s" % (seed, k)] = v
                meta_runs[str(seed)] = {"n_steps": len(rec["losses"]), "ckpt_steps": [h["global_step"] for h in hist]}
            print(name, "seed", seed, vname, "final ndcg@10 %.4f" % nd[-1, -1], "steps", len(rec["losses"]))
    out["meta"] = json.dumps({"name": name, "algo": algo_key, "class": cls_name, "seeds": list(seeds), "topn": list(CONV_TOPN), "dataset": dataset_meta,
                              "variants": [v[0] for v in variants], "settings": settings,
                              "argv": ["--batch_size", str(batch), "--max_train_iteration", str(n_iter),
                                       "--steps_per_checkpoint", str(ckpt_every)],
                              "param_keys": list(init_sd.keys()), "prop_keys": list(init_prop.keys()) if init_prop else [],
                              "runs": meta_runs})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    fin = np.asarray([[out["s%d_%s_ndcg" % (s, v[0])][-1, -1] for v in variants] for s in seeds])
    print("wrote", name, "final ndcg@10 per seed x variant:\n", np.round(fin, 4), "\nmean", fin.mean(0), "max spread", np.ptp(fin, axis=1).max())


def run_click_models_case(ultra, name, seed=71):
    """The reference's three click simulators (click_models.py:68-110 PBM, 113-186 UBM, 187-236 cascade) loaded from its own
    example JSONs, on seeded label lists of 5 / 10 / 17 documents (the last beyond the 10-entry examination tables): clicks,
    examination and click probabilities per position, and estimatePropensityWeightsForOneList on the sampled clicks - all
    drawn from Python's `random` stream after random.seed(seed)."""
    from ultra.utils import click_models as RCM
    files = {"pbm": "pbm_0.1_1.0_4_1.0.json", "ubm": "ubm_0.1_1_4_1.0.json", "cascade": "cascade_0.1_1.0_4_1.0.json"}
    rng = np.random.RandomState(seed)
    lists = [rng.randint(0, 5, size=n).tolist() for n in (5, 10, 17, 10, 17, 12)]
    out = {"meta": json.dumps({"seed": seed, "models": list(files), "files": files, "n_lists": len(lists)})}
    for i, lab in enumerate(lists):
        out["labels%d" % i] = np.asarray(lab, np.int64)
    for key, fn in files.items():
        model = RCM.loadModelFromJson(json.load(open(os.path.join(REF, "example", "ClickModel", fn))))
        random.seed(seed)
        for i, lab in enumerate(lists):
            c, e, p = model.sampleClicksForOneList(list(lab))
            out["%s_clicks%d" % (key, i)] = np.asarray(c, np.float64)
            out["%s_exam%d" % (key, i)] = np.asarray(e, np.float64)
            out["%s_cprob%d" % (key, i)] = np.asarray(p, np.float64)
            for flag in (False, True):
                # (integer clicks: the reference's `use_non_clicked_data | click_list[r] > 0` raises on the cascade model's 0.0 floats)
                out["%s_pw%d_%d" % (key, i, int(flag))] = np.asarray(model.estimatePropensityWeightsForOneList([int(x) for x in c], flag), np.float64)
        out["%s_next_uniform" % key] = np.float64(random.random())  # where the stream stands afterwards
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: int(out[k + "_clicks2"].sum()) for k in files})


def run_metrics_case(ultra, name, seed=61):
    """Every host metric the reference's factory registers (metrics.py:36-153), evaluated BY THE REFERENCE on seeded scores /
    labels - two shapes, labels with invalid (-1) entries and PAD-masked scores, ties, an all-irrelevant list."""
    from ultra.utils import metrics as RM
    RM.RankingMetricKey.MAX_LABEL = 4.0
    rng = np.random.RandomState(seed)
    out = {"meta": json.dumps({"topn": [1, 3, 5, 10], "max_label": 4.0, "keys": ["ndcg", "mrr", "err", "precision", "arp", "map", "ordered_pair_accuracy"],
                               "note": "dcg raises in the reference (gather on weights=None, metrics.py:191-221 called without "
                                       "weights at :519-523); precision returns ONE scalar whatever topn is (metrics.py:373-405)"})}
    for tag, (B, L) in (("a", (9, 12)), ("b", (5, 7))):
        y = rng.randint(0, 5, size=(B, L)).astype(np.float32)
        s = rng.normal(size=(B, L)).astype(np.float32)
        y[1, :] = 0.0                      # a list without a relevant document
        s[2, 3] = s[2, 4]                  # a tie
        y[3, -2:] = -1.0                   # invalid labels (metrics.py:251-264)
        s[4, -3:] = -100000.0              # PAD-masked scores (base_algorithm.py:88-116)
        y[4, -3:] = 0.0
        out[tag + "_labels"], out[tag + "_scores"] = y, s
        for key in ("ndcg", "mrr", "err", "precision", "arp", "map", "ordered_pair_accuracy"):
            fn = RM.make_ranking_metric_fn(key, [1, 3, 5, 10])
            out["%s_%s" % (tag, key)] = np.asarray(fn(torch.from_numpy(y), torch.from_numpy(s), None).detach().cpu().numpy(), np.float64).reshape(-1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: v for k, v in out.items() if k.startswith("a_") and not k.endswith(("labels", "scores"))})


CASES = {
    "metrics_host": lambda u: run_metrics_case(u, "metrics_host"),
    "feeds_toy": lambda u: run_feed_case(u, "feeds_toy"),
    "click_models": lambda u: run_click_models_case(u, "click_models"),
    # the driver itself (main.py) on the toy dataset: losses, checkpoint schedule, printed metrics, saved tensors
    "driver_toy": lambda u: run_driver_case(u, "driver_toy"),
    # end-of-training NDCG@10 of the reference's main.py with a click feed, 5 seeds x {1 thread, 8 threads, one-rounding init}
    "conv_ipw": lambda u: run_convergence_case(u, "conv_ipw", "ipw"),
    "conv_dla": lambda u: run_convergence_case(u, "conv_dla", "dla", n_ulp=6),
    "conv_pairdebias": lambda u: run_convergence_case(u, "conv_pairdebias", "pairdebias"),
    "conv_dla_synth": lambda u: run_convergence_case(u, "conv_dla_synth", "dla", n_ulp=6, synth=True, seeds=tuple(range(10))),
    # tiny, two teacher-forced steps each
    "na_tiny": lambda u: run_train_case(u, "na_tiny", "na", 136, 10, 8, [32, 16], 2, 11),
    "ipw_tiny": lambda u: run_train_case(u, "ipw_tiny", "ipw", 136, 10, 8, [32, 16], 2, 12),
    "dla_tiny": lambda u: run_train_case(u, "dla_tiny", "dla", 136, 10, 8, [32, 16], 2, 13),
    "pairdebias_tiny": lambda u: run_train_case(u, "pairdebias_tiny", "pairdebias", 136, 10, 8, [32, 16], 2, 14),
    "lambdarank_tiny": lambda u: run_train_case(u, "lambdarank_tiny", "lambdarank", 136, 10, 8, [32, 16], 2, 15),
    # ragged / odd shapes: nothing a multiple of 4, 16 or 64
    "ipw_odd": lambda u: run_train_case(u, "ipw_odd", "ipw", 7, 3, 5, [5], 2, 16, n_queries=16),
    "dla_odd": lambda u: run_train_case(u, "dla_odd", "dla", 13, 7, 9, [19, 6, 3], 2, 17, n_queries=32),
    "pairdebias_odd": lambda u: run_train_case(u, "pairdebias_odd", "pairdebias", 13, 7, 9, [19, 6, 3], 2, 18, n_queries=32),
    "lambdarank_odd": lambda u: run_train_case(u, "lambdarank_odd", "lambdarank", 13, 7, 9, [19, 6, 3], 2, 19, n_queries=32),
    # DLA with logits_to_prob=sigmoid (dla.py:21-22: sigmoid(x - mean(x))) and non-default loss weight / propensity lr
    "dla_sigmoid": lambda u: run_train_case(u, "dla_sigmoid", "dla", 136, 10, 8, [32, 16], 2, 25,
                                            algo_hparams="logits_to_prob=sigmoid"),
    "dla_sigmoid_odd": lambda u: run_train_case(u, "dla_sigmoid_odd", "dla", 13, 7, 9, [19, 6], 2, 26, n_queries=32,
                                                algo_hparams="logits_to_prob=sigmoid,ranker_loss_weight=0.5,"
                                                             "propensity_learning_rate=0.02"),
    # next row 8f.3: RegressionEM (uniforms of the Bernoulli draw recorded)
    "regem_tiny": lambda u: run_train_case(u, "regem_tiny", "regem", 136, 10, 8, [32, 16], 2, 23),
    "regem_odd": lambda u: run_train_case(u, "regem_odd", "regem", 13, 7, 9, [19, 6, 3], 2, 24, n_queries=32),
    # next row 8f.1: the SetRank ranking model (addressed as ultra.ranking_model.SetRank.SetRank) under IPW / NA
    "setrank_tiny": lambda u: run_train_case(u, "setrank_tiny", "ipw", 24, 10, 8, None, 2, 41,
                                             model_cls="ultra.ranking_model.SetRank.SetRank",
                                             model_extra="d_model=32,num_heads=4,num_layers=2,diff=16"),
    "setrank_odd": lambda u: run_train_case(u, "setrank_odd", "na", 13, 7, 5, None, 2, 42, n_queries=16,
                                            model_cls="ultra.ranking_model.SetRank.SetRank",
                                            model_extra="d_model=24,num_heads=3,num_layers=1,diff=12"),
    # the reference's own SetRank example pairs it with DLA (example/offline_setting/dla_exp_settings_setrank.json)
    "setrank_dla_tiny": lambda u: run_train_case(u, "setrank_dla_tiny", "dla", 24, 10, 8, None, 2, 44,
                                                 model_cls="ultra.ranking_model.SetRank.SetRank",
                                                 model_extra="d_model=32,num_heads=2,num_layers=1,diff=16"),
    # config-5 layer shapes (F220, L100, d_model 256, 8 heads, 2 layers, dff 64) at B = 2
    "setrank_cfg5_b2": lambda u: run_train_case(u, "setrank_cfg5_b2", "ipw", 220, 100, 2, None, 1, 43, n_queries=4,
                                                model_cls="ultra.ranking_model.SetRank.SetRank", model_extra=""),
    # k = 0 (the Linear ranking model: LayerNorm -> Linear(F,1))
    "na_linear": lambda u: run_train_case(u, "na_linear", "na", 136, 10, 8, None, 2, 20,
                                          model_cls="ultra.ranking_model.Linear"),
    # relu activation
    "ipw_relu": lambda u: run_train_case(u, "ipw_relu", "ipw", 24, 10, 8, [16, 8], 1, 21,
                                         model_extra=",activation_func=relu"),
    # the remaining activations of base_ranking_model.py:63-69
    "na_tanh": lambda u: run_train_case(u, "na_tanh", "na", 24, 10, 8, [16, 8], 2, 51, model_extra=",activation_func=tanh"),
    "na_sigmoid": lambda u: run_train_case(u, "na_sigmoid", "na", 24, 10, 8, [16, 8], 2, 52, model_extra=",activation_func=sigmoid"),
    # (activation_func=selu raises in the reference: ACT_FUNC_DIC holds a plain function, nn.Sequential.add_module rejects it)
    # l2_loss > 0: g += l2 * p and - every algorithm but DLA - the gradient clip silently skipped;
    # l2_loss = 1 makes ||l2 * p|| > max_gradient_norm, so a clip that was NOT skipped would show
    "ipw_l2": lambda u: run_train_case(u, "ipw_l2", "ipw", 24, 10, 8, [16, 8], 2, 54, algo_hparams="l2_loss=1.0"),
    "na_l2": lambda u: run_train_case(u, "na_l2", "na", 24, 10, 8, [16, 8], 2, 55, algo_hparams="l2_loss=1.0"),
    "dla_l2": lambda u: run_train_case(u, "dla_l2", "dla", 24, 10, 8, [16, 8], 1, 56, algo_hparams="l2_loss=1.0"),
    "pairdebias_l2": lambda u: run_train_case(u, "pairdebias_l2", "pairdebias", 24, 10, 8, [16, 8], 2, 57, algo_hparams="l2_loss=1.0"),
    "regem_l2": lambda u: run_train_case(u, "regem_l2", "regem", 24, 10, 8, [16, 8], 2, 58, algo_hparams="l2_loss=1.0"),
    # sgd strategy
    "ipw_sgd": lambda u: run_train_case(u, "ipw_sgd", "ipw", 24, 10, 8, [16, 8], 2, 22,
                                        algo_hparams="grad_strategy=sgd"),
    # one config-2-shaped step (BASELINE.json configs[1]): F136 L10 B256 DNN[256,256] IPW
    "ipw_cfg2": lambda u: run_train_case(u, "ipw_cfg2", "ipw", 136, 10, 256, [256, 256], 1, 2, n_queries=512),
    # validation with pads
    "valid_tiny": lambda u: run_valid_case(u, "valid_tiny", 136, 12, 8, [32, 16], 31, (2, 12)),
    "valid_odd": lambda u: run_valid_case(u, "valid_odd", 13, 37, 5, [19, 6, 3], 32, (2, 37), n_queries=13),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    torch.set_num_threads(1)  # bit-stable fixtures (reference is deterministic at a fixed thread count)
    ultra = import_reference()
    for name, fn in CASES.items():
        if args.only and args.only != name:
            continue
        fn(ultra)


if __name__ == "__main__":
    main()
