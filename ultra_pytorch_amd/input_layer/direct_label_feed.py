"""DirectLabelFeed — true relevance labels, random batches for training and sequential `get_next_batch` for
evaluation (reference direct_label_feed.py:22-284).  Note (Appendix A.13): get_batch draws exactly batch_size
queries and SKIPS all-zero lists without resampling, so it may return fewer than batch_size lists."""
import random

from ..utils import HParams
from .base_input_feed import BaseInputFeed


class DirectLabelFeed(BaseInputFeed):
    def __init__(self, model, batch_size, hparam_str):
        self.hparams = HParams(use_max_candidate_num=True)
        self.hparams.parse(hparam_str)
        self.start_index, self.count = 0, 1
        self.rank_list_size = model.max_candidate_num if self.hparams.use_max_candidate_num else model.rank_list_size
        self.feature_size, self.batch_size, self.model = model.feature_size, batch_size, model
        print("Create direct label feed with list size %d with feature size %d" % (self.rank_list_size, self.feature_size))

    def prepare_true_labels_with_index(self, data_set, index, docid_inputs, letor_features, labels, check_validation=True):
        label_list = [0 if data_set.initial_list[index][x] < 0 else data_set.labels[index][x] for x in range(self.rank_list_size)]
        if check_validation and sum(label_list) == 0:
            return
        self._add_list(data_set, index, label_list, docid_inputs, letor_features, labels)

    def get_batch(self, data_set, check_validation=False, data_format="ULTRA"):
        self._check(data_set)
        length = len(data_set.initial_list)
        docid_inputs, letor_features, labels, rank_list_idxs = [], [], [], []
        for _ in range(self.batch_size):
            i = int(random.random() * length)
            rank_list_idxs.append(i)
            self.prepare_true_labels_with_index(data_set, i, docid_inputs, letor_features, labels, check_validation)
        return self._assemble(docid_inputs, letor_features, labels), {
            "rank_list_idxs": rank_list_idxs, "input_list": docid_inputs, "click_list": labels, "letor_features": letor_features}

    def get_next_batch(self, index, data_set, check_validation=False, data_format="ULTRA"):
        self._check(data_set)
        docid_inputs, letor_features, labels = [], [], []
        for offset in range(min(self.batch_size, len(data_set.initial_list) - index)):
            self.prepare_true_labels_with_index(data_set, index + offset, docid_inputs, letor_features, labels, check_validation)
        return self._assemble(docid_inputs, letor_features, labels), {"input_list": docid_inputs, "click_list": labels}

    def get_data_by_index(self, data_set, index, check_validation=False):
        self._check(data_set)
        docid_inputs, letor_features, labels = [], [], []
        self.prepare_true_labels_with_index(data_set, index, docid_inputs, letor_features, labels, check_validation)
        return self._assemble(docid_inputs, letor_features, labels), {"input_list": docid_inputs, "click_list": labels}
