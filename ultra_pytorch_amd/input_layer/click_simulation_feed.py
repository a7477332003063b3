"""ClickSimulationFeed — uniform query sampling + simulated clicks (reference click_simulation_feed.py:26-294).
Random stream: one `random.random()` per query pick, then one per list position inside the click model, lists
without a click are rejected and resampled when check_validation is set (:89-91,125-131)."""
import json
import random

from ..utils import HParams
from ..utils import click_models as cm
from .base_input_feed import BaseInputFeed


class ClickSimulationFeed(BaseInputFeed):
    def __init__(self, model, batch_size, hparam_str):
        self.hparams = HParams(click_model_json="./example/ClickModel/pbm_0.1_1.0_4_1.0.json", oracle_mode=False,
                               dynamic_bias_eta_change=0.0, dynamic_bias_step_interval=1000)
        print("Create simluation feed")
        print(hparam_str)
        self.hparams.parse(hparam_str)
        self.click_model = None
        if not self.hparams.oracle_mode:
            import os
            path = self.hparams.click_model_json
            if not os.path.exists(path):  # the reference's default is relative to its repo root; same file ships here
                alt = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", os.path.basename(path))
                path = alt if os.path.exists(alt) else path
            with open(path) as fin:
                self.click_model = cm.loadModelFromJson(json.load(fin))
        self.start_index, self.count = 0, 1
        self.rank_list_size, self.feature_size = model.rank_list_size, model.feature_size
        self.batch_size, self.model = batch_size, model
        self.global_batch_count = 0

    def _labels_for(self, data_set, i):
        return [0 if data_set.initial_list[i][x] < 0 else data_set.labels[i][x] for x in range(self.rank_list_size)]

    def prepare_sim_clicks_with_index(self, data_set, index, docid_inputs, letor_features, labels, check_validation=True):
        label_list = self._labels_for(data_set, index)
        click_list = label_list if self.hparams.oracle_mode else self.click_model.sampleClicksForOneList(list(label_list))[0]
        if check_validation and sum(click_list) == 0:
            return
        self._add_list(data_set, index, click_list, docid_inputs, letor_features, labels)

    def get_batch(self, data_set, check_validation=False, data_format="ULTRA"):
        self._check(data_set)
        length = len(data_set.initial_list)
        docid_inputs, letor_features, labels, rank_list_idxs = [], [], [], []
        while len(docid_inputs) < self.batch_size:
            i = int(random.random() * length)
            before = len(docid_inputs)
            self.prepare_sim_clicks_with_index(data_set, i, docid_inputs, letor_features, labels, check_validation)
            if len(docid_inputs) > before:
                rank_list_idxs.append(i)
        input_feed = self._assemble(docid_inputs, letor_features, labels)
        info_map = {"rank_list_idxs": rank_list_idxs, "input_list": docid_inputs, "click_list": labels,
                    "letor_features": letor_features}
        self.global_batch_count += 1
        if self.hparams.dynamic_bias_eta_change != 0 and not self.hparams.oracle_mode:
            if self.global_batch_count % self.hparams.dynamic_bias_step_interval == 0:
                self.click_model.eta += self.hparams.dynamic_bias_eta_change
                self.click_model.setExamProb(self.click_model.eta)
        return input_feed, info_map

    def get_next_batch(self, index, data_set, check_validation=False, data_format="ULTRA"):
        self._check(data_set)
        docid_inputs, letor_features, labels = [], [], []
        for offset in range(min(self.batch_size, len(data_set.initial_list) - index)):
            self.prepare_sim_clicks_with_index(data_set, index + offset, docid_inputs, letor_features, labels, check_validation)
        return self._assemble(docid_inputs, letor_features, labels), {"input_list": docid_inputs, "click_list": labels}

    def get_data_by_index(self, data_set, index, check_validation=False):
        self._check(data_set)
        docid_inputs, letor_features, labels = [], [], []
        self.prepare_sim_clicks_with_index(data_set, index, docid_inputs, letor_features, labels, check_validation)
        return self._assemble(docid_inputs, letor_features, labels), {"input_list": docid_inputs, "click_list": labels}
