"""CPU restatement of the evaluation metrics (TEST INFRASTRUCTURE ONLY).

    cut_positions   run.py:137-142          k = argmax_j p[i, j] + 1 (first maximum)
    Metric.f1       utils/metrics.py:15-24  F1 of the top-k prefix, float64, batch mean
    Metric.dcg      utils/metrics.py:26-38  penalised DCG of the top-k prefix, float64, batch mean

Known answers carried by the reference (utils/metrics.py:104-109):
    labels [[1,0,1],[0,0,1],[1,0,0]], k_s [1,2,1] -> f1 0.5555555555555555, dcg 0.1230234154761809
"""
import math

import numpy as np

# utils/metrics.py:7 - python-float log with explicit base 2, 300 entries
DCG_COEF = [math.log(j + 2, 2) for j in range(300)]


def dcg_coef(seq_len):
    if seq_len > len(DCG_COEF):
        return [math.log(j + 2, 2) for j in range(seq_len)]
    return DCG_COEF[:seq_len]


def cut_positions(p):
    """p: (B,S) or (B,S,1) array of cut probabilities -> (B,) int k in 1..S."""
    p = np.asarray(p)
    if p.ndim == 3:
        p = p[:, :, 0]
    return np.argmax(p, axis=1) + 1


def f1_per_list(labels, k_s):
    labels = np.asarray(labels, dtype=np.float64)
    out = np.zeros(len(labels), dtype=np.float64)
    for i, k in enumerate(k_s):
        k = int(k)
        n_rel = labels[i].sum()
        hit = labels[i, :k].sum()
        prec = hit / k
        rec = hit / n_rel if n_rel != 0 else 0.0
        out[i] = 2 * prec * rec / (prec + rec) if prec + rec != 0 else 0.0
    return out


def dcg_per_list(labels, k_s, penalty=-1):
    labels = np.asarray(labels)
    out = np.zeros(len(labels), dtype=np.float64)
    for i, k in enumerate(k_s):
        k = int(k)
        head = labels[i, :k]
        coef = np.asarray(dcg_coef(labels.shape[1])[:k], dtype=np.float64)
        gain = np.where(head == 1, 1.0, float(penalty))
        out[i] = (gain / coef).sum()
    return out


class Metric:
    @classmethod
    def f1(cls, labels, k_s):
        return float(np.mean(f1_per_list(labels, k_s)))

    @classmethod
    def dcg(cls, labels, k_s, penalty=-1):
        return float(np.mean(dcg_per_list(labels, k_s, penalty)))


def bicut_cut_positions(output):
    """run.py:131-136: argmax over the two classes per position; k = S when every position says continue (1), else the
    first truncate position + 1.  `output` numpy (B,S,2)."""
    pred = np.argmax(output, axis=2)
    S = pred.shape[1]
    return np.array([S if r.sum() == S else int(np.argmin(r)) + 1 for r in pred], dtype=np.int64)


def taskr_metric(labels, predictions):
    """utils/metrics.py:40-57: per list, documents re-ordered by descending prediction; relevant documents earn
    +1/log2(i+2) at sorted position i, the others -1/log2(i+2); mean over the batch."""
    labels, predictions = np.asarray(labels), np.asarray(predictions)
    out = []
    for pred, lab in zip(predictions, labels):
        order = np.argsort(-pred, kind="stable")
        gain = np.where(lab[order] != 0, 1.0, -1.0) / np.log2(np.arange(len(order)) + 2.0)
        out.append(gain.sum())
    return float(np.mean(out))


def taskc_metric(labels, predictions):
    """utils/metrics.py:59-76: mean over the lists holding both classes of the ROC AUC (pairs of a positive and a
    negative document ranked correctly, ties counting 1/2 - what sklearn's roc_auc_score computes)."""
    labels, predictions = np.asarray(labels), np.asarray(predictions, dtype=np.float64)
    total, count = 0.0, 0
    for pred, lab in zip(predictions, labels):
        pos, neg = pred[lab != 0], pred[lab == 0]
        if len(pos) == 0 or len(neg) == 0:
            continue
        diff = pos[:, None] - neg[None, :]
        total += ((diff > 0).sum() + 0.5 * (diff == 0).sum()) / (len(pos) * len(neg))
        count += 1
    return total / count
