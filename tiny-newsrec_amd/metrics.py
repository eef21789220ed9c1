"""Ranking metrics of the eval path (AUC, MRR, nDCG@k, CTR@k) in numpy on the host, as in the reference
(Tiny-NewsRec/metrics.py:5-29; eval only).  All of them are functions of the labels
sorted by descending score, so they share one helper; AUC is the Mann-Whitney statistic with average ranks on
ties (= sklearn.metrics.roc_auc_score, which the reference imports) so that eval needs no sklearn."""
import numpy as np


def _by_score(labels, scores, k=None):
    """labels reordered by descending score (ties: later index first, like argsort()[::-1]), cut at k."""
    top = np.argsort(scores)[::-1]
    return np.take(labels, top if k is None else top[:k])


def _dcg(ranked_labels):
    positions = np.arange(len(ranked_labels))
    return float(((2.0 ** ranked_labels - 1.0) / np.log2(positions + 2)).sum())


def ndcg_score(y_true, y_score, k=10):
    ideal = _dcg(_by_score(y_true, y_true, k))
    return _dcg(_by_score(y_true, y_score, k)) / ideal


def dcg_score(y_true, y_score, k=10):
    return _dcg(_by_score(y_true, y_score, k))


def mrr_score(y_true, y_score):
    ranked = _by_score(y_true, y_score)
    reciprocal = 1.0 / np.arange(1, len(ranked) + 1)
    return float((ranked * reciprocal).sum() / ranked.sum())


def ctr_score(y_true, y_score, k=1):
    return float(_by_score(y_true, y_score, k).mean())


def roc_auc_score(y_true, y_score):
    labels = np.asarray(y_true)
    vals = np.asarray(y_score, dtype=np.float64)
    asc = np.argsort(vals, kind="mergesort")
    sorted_vals = vals[asc]
    rank = np.empty(len(vals), dtype=np.float64)
    start = 0
    while start < len(vals):                       # one tie group at a time -> mean rank of the group
        stop = start
        while stop + 1 < len(vals) and sorted_vals[stop + 1] == sorted_vals[start]:
            stop += 1
        rank[asc[start:stop + 1]] = (start + stop) / 2.0 + 1.0
        start = stop + 1
    is_pos = labels == 1
    p, q = int(is_pos.sum()), int((~is_pos).sum())
    return float((rank[is_pos].sum() - p * (p + 1) / 2.0) / (p * q))
