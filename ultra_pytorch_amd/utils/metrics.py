"""Host-side ranking metrics with the reference's factory interface (ultra/utils/metrics.py:36-153).

NDCG on the validation path runs as a HIP kernel (ultr_ndcg); the other metrics that appear in the example
settings (mrr, err) and the rest of the reference's factory table (arp, precision, map, ordered_pair_accuracy, dcg) are
evaluated here on host tensors — SURVEY.md §8 a13 marks them "host restatement suffices".  weights=None semantics only
(what every validation() call passes).  Pinned to the reference's own outputs: tests/golden/metrics_host.npz.
Two places where this module is usable and the reference is not: `dcg` (the reference's raises: it gathers a `weights` of None,
metrics.py:519-523 -> :191-221) and `precision` (the reference returns ONE scalar whatever topn is, metrics.py:373-405, which
validation()'s zip over metrics_topn cannot iterate; here the same value is repeated per cut-off).
"""
import numpy as np
import torch


class RankingMetricKey(object):
    MRR, ERR, ARP, NDCG, DCG, PRECISION = "mrr", "err", "arp", "ndcg", "dcg", "precision"
    MAP, ORDERED_PAIR_ACCURACY = "map", "ordered_pair_accuracy"  # metrics.py:56, 59
    MAX_LABEL = None  # set by the data loader from settings.json (data_utils.py:96)


def _safe_div(n, d):
    return torch.where(torch.eq(d, 0), torch.zeros_like(n), torch.div(n, d))


def _prepare(labels, predictions, topn):
    """metrics.py:224-265: clip topn to the list size; invalid labels (< 0) -> label 0, score rowmin - 1e-6."""
    L = predictions.shape[1]
    topn = [L] if topn is None else [min(int(n), L) for n in topn]
    ok = labels >= 0.0
    labels = torch.where(ok, labels, torch.zeros_like(labels))
    predictions = torch.where(ok, predictions, torch.min(predictions, dim=1, keepdim=True).values - 1e-6)
    return labels, predictions, topn


def _sorted_labels(labels, predictions):
    idx = torch.argsort(predictions, dim=-1, descending=True, stable=True)
    return torch.gather(labels, 1, idx)


def _dcg(predictions, labels, topn):
    L = labels.shape[1]
    sl = _sorted_labels(labels, predictions).float()
    disc = 1.0 / torch.log2(torch.arange(L, dtype=torch.float32) + 2.0)
    cum = torch.cumsum(((torch.pow(torch.tensor(2.0), sl) - 1.0) * disc)[:, :max(topn)], dim=1)
    return cum[:, torch.tensor(topn, dtype=torch.long) - 1]


def normalized_discounted_cumulative_gain(labels, predictions, weights=None, topn=None, name=None):
    labels, predictions, topn = _prepare(labels, predictions, topn)
    return torch.mean(_safe_div(_dcg(predictions, labels, topn), _dcg(labels, labels, topn)), dim=0)


def discounted_cumulative_gain(labels, predictions, weights=None, topn=None, name=None):
    labels, predictions, topn = _prepare(labels, predictions, topn)
    return torch.mean(_dcg(predictions, labels, topn), dim=0)


def mean_reciprocal_rank(labels, predictions, weights=None, topn=None, name=None):
    L = predictions.shape[-1]
    labels, predictions, topn = _prepare(labels, predictions, topn)
    rel = torch.ge(_sorted_labels(labels, predictions), 1.0).float()
    rr = 1.0 / torch.arange(1, L + 1, dtype=torch.float32)
    return torch.mean(torch.max(rel * rr, dim=1).values).repeat(len(topn))


def expected_reciprocal_rank(labels, predictions, weights=None, topn=None, name=None):
    labels, predictions, topn = _prepare(labels, predictions, topn)
    sl = _sorted_labels(labels, predictions).float()
    L = sl.shape[-1]
    max_label = 4.0 if RankingMetricKey.MAX_LABEL is None else float(RankingMetricKey.MAX_LABEL)
    rel = (torch.pow(torch.tensor(2.0), sl) - 1) / (2.0 ** max_label)
    non_rel = torch.cumprod(1.0 - rel, dim=1) / (1.0 - rel)
    rr = 1.0 / torch.arange(1, L + 1, dtype=torch.float32)
    out = [torch.mean(torch.sum(rel * non_rel * rr * torch.ge(rr, 1.0 / n).float(), dim=1)) for n in topn]
    return torch.stack(out)


def _per_list_relevance_weight(labels):
    """_per_example_weights_to_per_list_weights with unit weights (metrics.py:173-188): 1 for a list with a relevant document, else 0."""
    rel = torch.ge(labels, 1.0).float()
    return _safe_div(torch.sum(rel, 1, keepdim=True), torch.sum(rel, 1, keepdim=True))


def precision(labels, predictions, weights=None, topn=None, name=None):
    """metrics.py:373-405: the share of relevant documents in the WHOLE list (topn is not applied there), zero for lists
    without one; a batch mean - repeated per cut-off here (module docstring)."""
    labels, predictions, topn = _prepare(labels, predictions, topn)
    rel = torch.ge(_sorted_labels(labels, predictions), 1.0).float()
    per_list = torch.sum(rel, 1, keepdim=True) / float(rel.shape[1])
    return torch.mean(per_list * _per_list_relevance_weight(labels)).repeat(len(topn))


def average_relevance_position(labels, predictions, weights=None, topn=None, name=None):
    """metrics.py:338-370: mean over lists of sum(position x label) / sum(label) over the whole ranked list (0 for a list of
    zeros); topn only sets the length of the (repeated) result."""
    labels, predictions, topn = _prepare(labels, predictions, topn)
    sl = _sorted_labels(labels, predictions).float()
    pos = torch.arange(1, sl.shape[-1] + 1, dtype=torch.float32)
    per_list = _safe_div(torch.sum(pos * sl, dim=1, keepdim=True), torch.sum(sl, dim=1, keepdim=True))
    return torch.mean(per_list).repeat(len(topn))


def mean_average_precision(labels, predictions, weights=None, topn=None, name=None):
    """metrics.py:408-453: per list, the mean over its relevant documents (label >= 1) of precision-at-their-rank; lists without a
    relevant document count as 0; batch mean, repeated per cut-off."""
    labels, predictions, topn = _prepare(labels, predictions, topn)
    rel = torch.ge(_sorted_labels(labels, predictions), 1.0).float()
    prec_at = torch.cumsum(rel, dim=1) / torch.arange(1, rel.shape[1] + 1, dtype=torch.float32)
    per_list = torch.nan_to_num(torch.sum(prec_at * rel, 1, keepdim=True) / torch.sum(rel, 1, keepdim=True))
    return torch.mean(per_list * _per_list_relevance_weight(labels)).repeat(len(topn))


def ordered_pair_accuracy(labels, predictions, weights=None, topn=None, name=None):
    """metrics.py:531-568: pairs (i, j) of valid documents with label_i > label_j AND score_i > score_j, as a MEAN OVER ALL
    batch x L x L ordered pairs (not over the pairs that differ in label - the reference's normalisation); repeated per cut-off."""
    clean, predictions, topn = _prepare(labels, predictions, topn)
    valid = torch.eq(clean, labels)
    vp = (valid.unsqueeze(2) & valid.unsqueeze(1)).float()
    dl = clean.unsqueeze(2) - clean.unsqueeze(1)
    dp = predictions.unsqueeze(2) - predictions.unsqueeze(1)
    gt = torch.gt(dl, 0).float()
    return torch.mean(gt * torch.gt(dp, 0).float() * gt * vp).repeat(len(topn))


def make_ranking_metric_fn(metric_key, topn=None, name=None):
    """Factory with the reference's signature: fn(labels, predictions, weights) -> tensor[len(topn)]."""
    table = {
        RankingMetricKey.NDCG: normalized_discounted_cumulative_gain,
        RankingMetricKey.DCG: discounted_cumulative_gain,
        RankingMetricKey.MRR: mean_reciprocal_rank,
        RankingMetricKey.ERR: expected_reciprocal_rank,
        RankingMetricKey.PRECISION: precision,
        RankingMetricKey.ARP: average_relevance_position,
        RankingMetricKey.MAP: mean_average_precision,
        RankingMetricKey.ORDERED_PAIR_ACCURACY: ordered_pair_accuracy,
    }
    assert metric_key in table, "metric_key %s not supported." % metric_key
    fn = table[metric_key]
    return lambda labels, predictions, weights=None: fn(labels, predictions, weights, topn=topn, name=name)
