"""Host-side ranking metrics with the reference's factory interface (ultra/utils/metrics.py:36-153).

NDCG on the validation path runs as a HIP kernel (ultr_ndcg); the other metrics that appear in the example
settings (mrr, err, dcg, precision, arp) are evaluated here on host tensors — SURVEY.md §8 a13 marks them
"host restatement suffices".  weights=None semantics only (what every validation() call passes).
"""
import numpy as np
import torch


class RankingMetricKey(object):
    MRR, ERR, ARP, NDCG, DCG, PRECISION = "mrr", "err", "arp", "ndcg", "dcg", "precision"
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


def precision(labels, predictions, weights=None, topn=None, name=None):
    labels, predictions, topn = _prepare(labels, predictions, topn)
    rel = torch.ge(_sorted_labels(labels, predictions), 1.0).float()
    return torch.stack([torch.mean(torch.sum(rel[:, :n], dim=1) / float(n)) for n in topn])


def average_relevance_position(labels, predictions, weights=None, topn=None, name=None):
    labels, predictions, topn = _prepare(labels, predictions, topn)
    sl = _sorted_labels(labels, predictions).float()
    L = sl.shape[-1]
    pos = torch.arange(1, L + 1, dtype=torch.float32)
    out = []
    for n in topn:
        m = (pos <= n).float()
        out.append(torch.sum(_safe_div(torch.sum(pos * sl * m, 1), torch.sum(sl * m, 1)) * torch.sum(sl * m, 1)) /
                   torch.clamp(torch.sum(sl * m), min=1e-12))
    return torch.stack(out)


def make_ranking_metric_fn(metric_key, topn=None, name=None):
    """Factory with the reference's signature: fn(labels, predictions, weights) -> tensor[len(topn)]."""
    table = {
        RankingMetricKey.NDCG: normalized_discounted_cumulative_gain,
        RankingMetricKey.DCG: discounted_cumulative_gain,
        RankingMetricKey.MRR: mean_reciprocal_rank,
        RankingMetricKey.ERR: expected_reciprocal_rank,
        RankingMetricKey.PRECISION: precision,
        RankingMetricKey.ARP: average_relevance_position,
    }
    assert metric_key in table, "metric_key %s not supported." % metric_key
    fn = table[metric_key]
    return lambda labels, predictions, weights=None: fn(labels, predictions, weights, topn=topn, name=name)
