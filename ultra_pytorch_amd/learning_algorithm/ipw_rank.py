"""IPWrank — inverse propensity weighting with a pre-estimated propensity table.
Drop-in for ultra.learning_algorithm.IPWrank (reference ipw_rank.py:31-211)."""
import collections.abc
import json
import os

import numpy as np
import torch

from ..utils import HParams
from .base_algorithm import BaseAlgorithm


class _LazyColumn(collections.abc.Sequence):
    """`propensity_weights{l}` as the reference leaves it in the feed (a Python list of B floats, ipw_rank.py:118-128),
    materialised on first use: nothing in main.py reads these entries, and ten `.tolist()` calls per step were a third of the
    Python time of `train` at config 2."""
    __slots__ = ("_owner", "_l", "_list")

    def __init__(self, owner, l):
        self._owner, self._l, self._list = owner, l, None

    def tolist(self):
        if self._list is None:
            self._list = self._owner.matrix()[self._l].tolist()
        return self._list

    def __getitem__(self, i):
        return self.tolist()[i]

    def __len__(self):
        return self._owner.batch

    def __eq__(self, other):
        return self.tolist() == (other.tolist() if isinstance(other, _LazyColumn) else other)

    def __repr__(self):
        return repr(self.tolist())


class _LazyWeights(object):
    """pw[l, b] = click[l, b] > 0 ? IPW_list[min(l, len - 1)] : 0 (propensity_estimator.py:22-42), computed when first read."""
    __slots__ = ("clicks", "table", "batch", "_pw")

    def __init__(self, clicks, table):
        self.clicks, self.table, self.batch, self._pw = clicks, table, int(clicks.shape[1]), None

    def matrix(self):  # [L, B]
        if self._pw is None:
            self._pw = np.where(self.clicks > 0, self.table[:, None], 0.0)
        return self._pw


class IPWrank(BaseAlgorithm):
    ENGINE_ALGO = "softmax"

    def __init__(self, data_set, exp_settings):
        self.hparams = HParams(
            propensity_estimator_type="ultra.utils.propensity_estimator.RandomizedPropensityEstimator",
            propensity_estimator_json="./example/PropensityEstimator/randomized_pbm_0.1_1.0_4_1.0.json",
            learning_rate=0.05, max_gradient_norm=5.0, loss_func="softmax_loss", l2_loss=0.0, grad_strategy="ada")
        print(exp_settings["learning_algorithm_hparams"])
        self.hparams.parse(exp_settings["learning_algorithm_hparams"])
        self._check_hparams()
        self._setup(data_set, exp_settings)
        # BasicPropensityEstimator.loadEstimatorFromFile: the JSON's "IPW_list" (propensity_estimator.py:44-56)
        path = self.hparams.propensity_estimator_json
        if not os.path.exists(path):  # the reference resolves its default relative to the repo root; we ship the same table
            alt = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", os.path.basename(path))
            path = alt if os.path.exists(alt) else path
        with open(path) as fin:
            self.IPW_list = [float(x) for x in json.load(fin)["IPW_list"]]
        self.ipw_table = torch.tensor(self.IPW_list, dtype=torch.float32, device=self.cuda)
        self._pw_table, self._lazy_pw = None, None
        self._pw_names = ["propensity_weights{0}".format(l) for l in range(self.max_candidate_num)]

    @property
    def propensity_weights(self):
        """[B, L], what the reference keeps as a list of per-list weight lists (ipw_rank.py:130)."""
        return None if self._lazy_pw is None else self._lazy_pw.matrix().T

    def train(self, input_feed):
        """ipw_rank.py:102-182.  The per-list Python loop over getPropensityForOneList is folded into the loss
        kernel (pw = click > 0 ? IPW_list[min(l, len-1)] : 0); the feed still gets the `propensity_weights{l}`
        entries the reference adds (ipw_rank.py:118-128)."""
        self.global_step += 1
        if not self.model.training:  # (nn.Module.train() walks every submodule: ~10 us a 47 us step does not have)
            self.model.train()
        L = self.rank_list_size
        clicks = self.create_input_feed(input_feed, L)  # [L, B] host (None for a device feed)
        if clicks is not None:
            if self._pw_table is None or self._pw_table.shape[0] != L:
                self._pw_table = np.asarray([self.IPW_list[min(l, len(self.IPW_list) - 1)] for l in range(L)])
            self._lazy_pw = lazy = _LazyWeights(clicks.copy(), self._pw_table)  # own copy: `clicks` is a view of the staging buffer
            for l in range(L):
                input_feed[self._pw_names[l]] = _LazyColumn(lazy, l)
        eng = self._train_engine(self.batch_size, L)
        sc = eng.train_step(self.model.flat_params, self.state_sum, self.letor_features, self.n_docs, self.docid_inputs,
                            self.labels_LB, ipw_table=self.ipw_table)
        self.loss = eng.read_loss()
        print(" Loss %f at Global Step %d: " % (self.loss, self.global_step))
        return self.loss, None, self.train_summary
