"""Dataset container + file formats of the ULTRA toolbox (host side, one-time I/O).

`Raw_data` keeps the reference's field names (features, dids, qids, initial_list, labels, initial_list_lengths,
rank_list_size, feature_size) because the feeds read them.  Formats: ULTRA (`<prefix>.feature/.init_list/.labels`)
and libsvm (`<prefix>.txt`), reference ultra/utils/data_utils.py:99-365; queries with < 2 documents or no relevant
document are dropped (:367-422); `pad` appends the all-zero PAD feature row and -1 list entries (:476-498).
"""
import json
import os

import numpy as np

from . import metrics


class Raw_data(object):
    def __init__(self, data_path=None, file_prefix=None, click_model_dir=None, rank_cut=None):
        self.data_path, self.file_prefix = data_path, file_prefix
        self.feature_size, self.rank_list_size = -1, -1
        self.removed_feature_ids = []
        self.features, self.dids, self.qids = [], [], []
        self.initial_list, self.labels, self.initial_scores, self.initial_list_lengths = [], [], [], []
        if data_path is None:
            return
        base = os.path.join(data_path, file_prefix, file_prefix)
        if os.path.isfile(base + ".feature"):
            if click_model_dir is not None:
                raise NotImplementedError("ULTRE (real click log) format is outside the scope of this engine")
            self._load_ultra(data_path, base, rank_cut)
        elif os.path.isfile(base + ".txt"):
            self._load_libsvm(data_path, base, file_prefix, rank_cut)
        else:
            raise IOError("no %s.feature or %s.txt" % (base, base))
        print("Finished reading %d queries with lists." % len(self.qids))
        assert self.feature_size > 0 and len(self.qids) > 0 and len(self.dids) > 0

    # ---- loaders -------------------------------------------------------------------------------------
    def _load_settings(self, data_path):
        settings = json.load(open(os.path.join(data_path, "settings.json")))
        self.feature_size = int(settings["feature_size"])
        if "removed_feature_ids" in settings:
            self.removed_feature_ids = sorted((i for i in settings["removed_feature_ids"] if i <= self.feature_size), reverse=True)
        metrics.RankingMetricKey.MAX_LABEL = settings["max_label"]

    def _parse_sparse(self, tokens):
        row = [0.0] * self.feature_size
        for tok in tokens:
            idx, val = tok.split(":")
            i = int(idx) - 1
            if i < self.feature_size:
                row[i] = float(val)
        for rf in self.removed_feature_ids:
            del row[rf - 1]
        return row

    def _load_ultra(self, data_path, base, rank_cut):
        self._load_settings(data_path)
        with open(base + ".feature") as fin:
            for line in fin:
                arr = line.strip().split(" ")
                self.dids.append(arr[0])
                self.features.append(self._parse_sparse(arr[1:]))
        self.feature_size -= len(self.removed_feature_ids)
        with open(base + ".init_list") as fin:
            for line in fin:
                arr = line.strip().split(" ")
                self.qids.append(arr[0])
                ids = [int(x) for x in arr[1:]]
                self.initial_list.append(ids[:rank_cut] if rank_cut else ids)
        self.rank_list_size = max(len(x) for x in self.initial_list)
        with open(base + ".labels") as fin:
            for line in fin:
                self.labels.append([float(x) for x in line.strip().split(" ")[1:][:self.rank_list_size]])
        self._finish()

    def _load_libsvm(self, data_path, base, file_prefix, rank_cut):
        self._load_settings(data_path)
        index = {}
        with open(base + ".txt") as fin:
            for line_num, line in enumerate(fin):
                arr = line.strip().split(" ")
                qid = arr[1].split(":")[1]
                if qid not in index:
                    index[qid] = len(index)
                    self.qids.append(qid)
                    self.initial_list.append([])
                    self.labels.append([])
                q = index[qid]
                if rank_cut and len(self.initial_list[q]) >= rank_cut:
                    continue
                # the reference stores line_num as the document index although skipped (rank_cut) lines do not
                # get a feature row; without rank_cut the two coincide.  We index the row actually stored.
                self.initial_list[q].append(len(self.features))
                self.labels[q].append(int(arr[0]))
                self.dids.append("%s_%s_%d" % (file_prefix, qid, line_num))
                self.features.append(self._parse_sparse(arr[2:]))
        self.feature_size -= len(self.removed_feature_ids)
        self.rank_list_size = max(len(x) for x in self.initial_list)  # over ALL queries, as the reference (:357-361)
        self._finish()

    def _finish(self):
        keep = [i for i in range(len(self.qids)) if len(self.initial_list[i]) >= 2 and sum(self.labels[i]) > 0]
        print("Remove %d invalid queries." % (len(self.qids) - len(keep)))
        self.qids = [self.qids[i] for i in keep]
        self.initial_list = [self.initial_list[i] for i in keep]
        self.labels = [self.labels[i] for i in keep]
        self.initial_list_lengths = [len(x) for x in self.initial_list]
        self.rank_list_size = max([self.rank_list_size] + self.initial_list_lengths)

    # ---- padding ---------------------------------------------------------------------------------------
    def pad(self, rank_list_size, pad_tails=True):
        self.rank_list_size = rank_list_size
        self.features.append([0.0] * self.feature_size)  # the PAD vector
        for i, lst in enumerate(self.initial_list):
            if len(lst) < rank_list_size:
                fill = [-1] * (rank_list_size - len(lst))
                self.initial_list[i] = lst + fill if pad_tails else fill + lst


def read_data(data_path, file_prefix, click_model_dir=None, rank_cut=None):
    return Raw_data(data_path, file_prefix, click_model_dir, rank_cut)


def merge_Summary(summary_list, weights):
    """Batch-size weighted mean of per-batch metric dicts (data_utils.py:501-514)."""
    total, wsum = {}, {}
    for summary, w in zip(summary_list, weights):
        for k, v in summary.items():
            total[k] = total.get(k, 0.0) + v * w
            wsum[k] = wsum.get(k, 0.0) + w
    return {k: total[k] / max(0.0000001, wsum[k]) for k in total}


def generate_ranklist_by_scores(data, rerank_scores):
    """qid -> [(did, score)] by descending score over the LIST POSITIONS, pads (doc index < 0) dropped; a score matrix
    whose shape does not match the initial lists raises, as the reference does (data_utils.py:567-617)."""
    if len(rerank_scores) != len(data.initial_list):
        raise ValueError("Rerank ranklists number must be equal to the initial list, %d != %d."
                         % (len(rerank_scores), len(data.initial_list)))
    out = {}
    for i, qid in enumerate(data.qids):
        scores = [float(s) for s in rerank_scores[i]]
        if len(scores) != len(data.initial_list[i]):
            raise ValueError("Rerank ranklists length must be equal to the gold list, %d != %d."
                             % (len(scores), len(data.initial_list[i])))
        order = sorted(range(len(scores)), key=lambda k: scores[k], reverse=True)  # a permutation: every position once
        out[qid] = [(data.dids[data.initial_list[i][k]], scores[k]) for k in order if data.initial_list[i][k] >= 0]
    return out


def output_ranklist(data, rerank_scores, output_path, file_name="test"):
    """TREC run file `<output_path><file_name>.ranklist` (data_utils.py:619-640); scores are written as plain
    floats (the reference prints `tensor(x)` because it passes tensor rows, Appendix A.14)."""
    ranklists = generate_ranklist_by_scores(data, rerank_scores)
    with open(os.path.join(output_path, file_name + ".ranklist"), "w") as fout:
        for qid in data.qids:
            for i, (did, score) in enumerate(ranklists[qid]):
                fout.write("%s Q0 %s %d %s Model\n" % (qid, did, i + 1, repr(score)))


def write_ultra_dataset(path, prefix, features, lists, labels, qids=None):
    """Write a dataset in ULTRA format (used by tests / synthetic data): features [n_docs, F] array, lists = list of
    doc-index lists, labels = list of label lists."""
    d = os.path.join(path, prefix)
    os.makedirs(d, exist_ok=True)
    base = os.path.join(d, prefix)
    with open(base + ".feature", "w") as f:
        for i, row in enumerate(np.asarray(features)):
            f.write("%s_%d %s\n" % (prefix, i, " ".join("%d:%.6f" % (j + 1, v) for j, v in enumerate(row) if v != 0.0)))
    with open(base + ".init_list", "w") as f:
        for q, lst in enumerate(lists):
            f.write("%s %s\n" % (qids[q] if qids else q, " ".join(str(x) for x in lst)))
    with open(base + ".labels", "w") as f:
        for q, lab in enumerate(labels):
            f.write("%s %s\n" % (qids[q] if qids else q, " ".join(str(float(x)) for x in lab)))
