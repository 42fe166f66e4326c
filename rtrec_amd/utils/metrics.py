"""Ranking metrics for Recommender.evaluate.

Definitions follow rtrec.utils.metrics (/root/reference/rtrec/utils/metrics.py:5-313): every
metric is a function of the 0/1 relevance vector of the first k = min(len(ranked), size)
recommendations and of |ground truth|; compute_scores averages them over queries.  Here the
relevance vector is formed once per query and all nine figures are derived from it.
"""
from __future__ import annotations

from collections import defaultdict
from math import log2
from typing import Any, Dict, Iterable, List, Sequence, Tuple


def _relevance(ranked_list: Sequence[Any], ground_truth: Sequence[Any], size: int) -> List[int]:
    truth = set(ground_truth) if not isinstance(ground_truth, (set, frozenset)) else ground_truth
    return [1 if item in truth else 0 for item in ranked_list[:min(len(ranked_list), size)]]


def _query_metrics(ranked_list: Sequence[Any], ground_truth: Sequence[Any], size: int) -> Dict[str, float]:
    rel = _relevance(ranked_list, ground_truth, size)
    k, n_true, tp = len(rel), len(ground_truth), sum(rel)
    empty_truth = n_true == 0
    both_empty_score = 1.0 if not ranked_list else 0.0

    prec = both_empty_score if empty_truth else (tp / k if k else 0.0)
    rec = both_empty_score if empty_truth else tp / n_true
    if empty_truth and not ranked_list:
        f1 = 1.0
    else:
        f1 = 2 * prec * rec / (prec + rec) if (prec + rec) > 0 else 0.0

    dcg = sum(1.0 / log2(pos + 2) for pos, r in enumerate(rel) if r)
    idcg = sum(1.0 / log2(pos + 2) for pos in range(min(n_true, size)))
    first = next((pos for pos, r in enumerate(rel) if r), None)

    running, ap_sum, ordered_pairs = 0, 0.0, 0
    for pos, r in enumerate(rel):
        if r:
            running += 1
            ap_sum += running / (pos + 1)
        else:
            ordered_pairs += running          # every earlier hit outranks this miss
    if empty_truth:
        ap = auc = both_empty_score
    else:
        denom = min(n_true, size)
        ap = ap_sum / denom if denom else 0.0
        if not ranked_list or tp == 0:
            auc = 0.0
        elif tp == k:
            auc = 1.0
        else:
            auc = ordered_pairs / (tp * (k - tp))
    return {"precision": prec, "recall": rec, "f1": f1, "ndcg": dcg / idcg if idcg > 0 else 0.0,
            "hit_rate": 1.0 if tp else 0.0, "mrr": 0.0 if first is None else 1.0 / (first + 1),
            "map": ap, "tp": tp, "auc": auc}


def ndcg(ranked_list, ground_truth, recommend_size): return _query_metrics(ranked_list, ground_truth, recommend_size)["ndcg"]
def precision(ranked_list, ground_truth, recommend_size): return _query_metrics(ranked_list, ground_truth, recommend_size)["precision"]
def recall(ranked_list, ground_truth, recommend_size): return _query_metrics(ranked_list, ground_truth, recommend_size)["recall"]
def true_positives(ranked_list, ground_truth, recommend_size): return _query_metrics(ranked_list, ground_truth, recommend_size)["tp"]
def f1_score(ranked_list, ground_truth, recommend_size): return _query_metrics(ranked_list, ground_truth, recommend_size)["f1"]
def hit(ranked_list, ground_truth, recommend_size): return _query_metrics(ranked_list, ground_truth, recommend_size)["hit_rate"]
def reciprocal_rank(ranked_list, ground_truth, recommend_size): return _query_metrics(ranked_list, ground_truth, recommend_size)["mrr"]
def auc(ranked_list, ground_truth, recommend_size): return _query_metrics(ranked_list, ground_truth, recommend_size)["auc"]
def average_precision(ranked_list, ground_truth, recommend_size): return _query_metrics(ranked_list, ground_truth, recommend_size)["map"]


def _mean_over_queries(key: str, ranked_lists, ground_truths, size: int) -> float:
    vals = [_query_metrics(r, g, size)[key] for r, g in zip(ranked_lists, ground_truths)]
    return sum(vals) / len(vals) if vals else 0.0


def mrr(ranked_lists, ground_truths, recommend_size): return _mean_over_queries("mrr", ranked_lists, ground_truths, recommend_size)
def map_score(ranked_lists, ground_truths, recommend_size): return _mean_over_queries("map", ranked_lists, ground_truths, recommend_size)


def compute_scores(evaluation_pairs: Iterable[Tuple[List[Any], List[Any]]], recommend_size: int) -> Dict[str, float]:
    totals: Dict[str, float] = defaultdict(float)
    n = 0
    for ranked_list, ground_truth in evaluation_pairs:
        n += 1
        for name, value in _query_metrics(ranked_list, ground_truth, recommend_size).items():
            totals[name] += value
    if n == 0:
        return defaultdict(float)
    return {name: (int(total) if name == "tp" else total / n) for name, total in totals.items()}
