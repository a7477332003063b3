"""Shared batch assembly of the feeds (reference base_input_feed.py:10-93, click_simulation_feed.py:101-174)."""
import numpy as np


class BaseInputFeed(object):
    MAX_SAMPLE_ROUND_NUM = 100

    @staticmethod
    def preprocess_data(data_set, hparam_str, exp_settings):
        return

    def _check(self, data_set):
        if len(data_set.initial_list[0]) < self.rank_list_size:
            raise ValueError("Input ranklist length must be no less than the required list size, %d != %d."
                             % (len(data_set.initial_list[0]), self.rank_list_size))

    def _assemble(self, docid_inputs, letor_features, labels):
        """Position-major float32 arrays; -1 (pad) becomes n_docs, the id of the all-zero PAD row."""
        n_docs, L, B = len(letor_features), self.rank_list_size, len(docid_inputs)
        ids = np.asarray(docid_inputs, dtype=np.float32).reshape(B, L)
        ids[ids < 0] = n_docs
        lab = np.asarray(labels, dtype=np.float32).reshape(B, L)
        feed = {self.model.letor_features_name: np.array(letor_features)}
        for l in range(L):
            feed[self.model.docid_inputs_name[l]] = np.ascontiguousarray(ids[:, l])
            feed[self.model.labels_name[l]] = np.ascontiguousarray(lab[:, l])
        return feed

    def _add_list(self, data_set, i, label_list, docid_inputs, letor_features, labels):
        base, row = len(letor_features), data_set.initial_list[i]
        for x in range(self.rank_list_size):
            if row[x] >= 0:
                letor_features.append(data_set.features[row[x]])
        docid_inputs.append([-1 if row[x] < 0 else base + x for x in range(self.rank_list_size)])
        labels.append(label_list)
