"""IPWrank — inverse propensity weighting with a pre-estimated propensity table.
Drop-in for ultra.learning_algorithm.IPWrank (reference ipw_rank.py:31-211)."""
import json
import os

import numpy as np
import torch

from ..utils import HParams
from .base_algorithm import BaseAlgorithm


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

    def train(self, input_feed):
        """ipw_rank.py:102-182.  The per-list Python loop over getPropensityForOneList is folded into the loss
        kernel (pw = click > 0 ? IPW_list[min(l, len-1)] : 0); the feed still gets the `propensity_weights{l}`
        entries the reference adds (ipw_rank.py:118-128)."""
        self.global_step += 1
        self.model.train()
        L = self.rank_list_size
        clicks = self.create_input_feed(input_feed, L)  # [L, B] host (None for a device feed)
        if clicks is not None:
            table = np.asarray([self.IPW_list[l] if l < len(self.IPW_list) else self.IPW_list[-1] for l in range(L)])
            pw = np.where(clicks > 0, table[:, None], 0.0)
            for l in range(L):
                input_feed["propensity_weights{0}".format(l)] = pw[l].tolist()
            self.propensity_weights = pw.T
        eng = self._train_engine(self.batch_size, L)
        sc = eng.train_step(self.model.flat_params, self.state_sum, self.letor_features, self.n_docs, self.docid_inputs,
                            self.labels_LB, ipw_table=self.ipw_table)
        self.loss = float(sc[0].item())
        print(" Loss %f at Global Step %d: " % (self.loss, self.global_step))
        return self.loss, None, self.train_summary
