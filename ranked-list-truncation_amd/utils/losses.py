"""Reward losses on the HIP hot path - drop-in for the criteria of the reference's utils/losses.py
that are on the hot path: ChoopyLoss :48-68, AttnCutLoss :71-96, RerankLoss :99-141,
MtCutLoss :164-191, DivLoss :194-233, BiCutLoss :11-45 and WassDistLoss :236-311 (section 8f row N4).  Same constructor signatures; `criterion(output, labels)`
returns a 0-d tensor supporting .backward() and .item().

The reference builds its (B,S) reward matrix with B*S python calls of O(S) tensor ops; here the
reward matrix, its softmax, the divergence and d(loss)/d(output) come out of ONE kernel pass
(`rlt_reward_loss`), one ranked list per wavefront or per half wavefront.
"""
import torch
from torch import nn

from rlt_hip import native as N
from rlt_hip import ops


def _metric_code(metric):
    return N.METRIC_F1 if metric == 'f1' else N.METRIC_DCG      # anything else => DCG (utils/losses.py:218-225)


def _prep(output, labels):
    N.require_cuda(output, labels)
    if output.dim() != 3 or output.shape[2] != 1:
        raise ValueError(f"expected a (B,S,1) cut distribution, got {tuple(output.shape)}")
    return N.f32c(output), N.f32c(labels)


class ChoopyLoss(nn.Module):
    def __init__(self, metric: str = 'f1'):
        super().__init__()
        self.metric = metric

    def forward(self, output, labels):
        p, y = _prep(output, labels)
        return ops.RewardLossFn.apply(p, y, _metric_code(self.metric), N.LOSS_EXPECT, 1.0)

    def forward_with_metrics(self, output, labels):
        """(loss, k, [sum F1@k, sum DCG@k]) from one kernel pass (utils.metrics.Metric.step)."""
        p, y = _prep(output, labels)
        return ops.RewardLossFn.apply(p, y, _metric_code(self.metric), N.LOSS_EXPECT, 1.0, -1.0, True)


class AttnCutLoss(nn.Module):
    def __init__(self, metric: str = 'f1', tau: float = 0.95):
        super().__init__()
        self.metric, self.tau = metric, tau

    def forward(self, output, labels):
        p, y = _prep(output, labels)
        return ops.RewardLossFn.apply(p, y, _metric_code(self.metric), N.LOSS_CE, float(self.tau))

    def forward_with_metrics(self, output, labels):
        p, y = _prep(output, labels)
        return ops.RewardLossFn.apply(p, y, _metric_code(self.metric), N.LOSS_CE, float(self.tau), -1.0, True)


class DivLoss(nn.Module):
    def __init__(self, metric: str = 'f1', tau: float = 0.85, div_type: str = 'kl', augmented: bool = True):
        super().__init__()
        self.metric, self.div_type, self.augmented = metric, div_type, augmented
        self.tau = tau if augmented else 1.

    def forward(self, output, labels):
        p, y = _prep(output, labels)
        kind = N.LOSS_KL if self.div_type == 'kl' else N.LOSS_JS
        return ops.RewardLossFn.apply(p, y, _metric_code(self.metric), kind, float(self.tau))

    def forward_with_metrics(self, output, labels):
        p, y = _prep(output, labels)
        kind = N.LOSS_KL if self.div_type == 'kl' else N.LOSS_JS
        return ops.RewardLossFn.apply(p, y, _metric_code(self.metric), kind, float(self.tau), -1.0, True)


class RerankLoss(nn.Module):
    """Batch-wide hinge between mean irrelevant and mean relevant score.  A batch without
    positives (or without negatives) gives a zero loss with zero gradients (the reference's intent,
    utils/losses.py:138; under torch>=2 the reference itself raises there)."""

    def __init__(self, margin: float = 5e-4, reduction: str = 'mean'):
        super().__init__()
        self.margin, self.reduction = margin, reduction

    def forward(self, output, labels):
        s, y = _prep(output, labels)
        return ops.RerankLossFn.apply(s, y, float(self.margin))


class MtCutLoss(nn.Module):
    def __init__(self, metric: str = 'f1', rerank_weight: float = 0.5, classi_weight: float = 0.5,
                 num_tasks: float = 3):
        super().__init__()
        self.rerank_weight, self.classi_weight = rerank_weight, classi_weight
        # the reference registers this (unused) parameter too (utils/losses.py:173): kept for
        # state_dict / RNG-consumption compatibility
        self.weights = nn.Parameter(torch.randn(int(num_tasks)), requires_grad=True)
        self.cutloss = DivLoss(metric=metric, div_type='js', augmented=True)
        self.rerankloss = RerankLoss()
        self.num_tasks = num_tasks
        self.metric = metric

    def _apply(self, output, labels, with_metrics):
        y_class = y_rerank = None
        if self.num_tasks == 3:
            y_class, y_rerank, y_cut = output
        elif self.num_tasks == 2.1:
            y_class, y_cut = output
        else:
            y_rerank, y_cut = output
        p, y = _prep(y_cut, labels)
        rr = None if y_rerank is None else N.f32c(y_rerank)
        cl = None if y_class is None else N.f32c(y_class)
        return ops.MtCutLossFn.apply(p, rr, cl, y, _metric_code(self.metric), float(self.cutloss.tau),
                                     float(self.rerank_weight), float(self.classi_weight),
                                     float(self.rerankloss.margin), with_metrics)

    def forward(self, output, labels):
        return self._apply(output, labels, False)

    def forward_with_metrics(self, output, labels):
        return self._apply(output, labels, True)


class BiCutLoss(nn.Module):
    """utils/losses.py:11-45 on the (B,S,2) output of BiCut; the position mask, the label-dependent reward pairs, the sum
    and d(loss)/d(output) come out of one kernel pass (`rlt_bicut_loss`)."""

    def __init__(self, alpha: float = 0.65, r: float = 0.0971134020, metric: str = 'nci'):
        super().__init__()
        self.metric, self.alpha, self.r = metric, alpha, r

    def forward(self, output, labels):
        N.require_cuda(output, labels)
        if output.dim() != 3 or output.shape[2] != 2:
            raise ValueError(f"expected the (B,S,2) output of BiCut, got {tuple(output.shape)}")
        return ops.BiCutLossFn.apply(N.f32c(output), N.f32c(labels), self.metric == 'nci', float(self.alpha), float(self.r))


class WassDistLoss(nn.Module):
    """utils/losses.py:236-311 (Sinkhorn distance between the batch's cut distributions and its label vectors); `metric`
    and `tau` are accepted and unused, as in the reference."""

    def __init__(self, eps: float = 1e-3, max_iter: int = 100, metric: str = 'f1', tau: float = 0.95, reduction='mean'):
        super().__init__()
        self.eps, self.max_iter, self.reduction = eps, max_iter, reduction     # one (B,B) problem: mean == sum

    def forward(self, output, labels):
        p, y = _prep(output, labels)
        B, S = y.shape
        return ops.WassDistLossFn.apply(p.reshape(B, S), y, float(self.eps), int(self.max_iter), 1e-1)
