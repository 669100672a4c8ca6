"""Cut-position metrics on the HIP hot path - drop-in for the reference's utils/metrics.py
(`Metric` :9-38, `Metric_for_Loss` :79-101).

`Metric.f1 / Metric.dcg` keep the reference's host interface (numpy labels (B,S), k_s (B,) ->
python float) but evaluate on the GPU with `rlt_cut_metrics`; `Metric.evaluate` is the
device-resident form the training loop uses (no host round trip of the (B,S) outputs).
"""
import numpy as np
import torch

from rlt_hip import native as N
from rlt_hip import ops


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("utils.metrics runs on the GPU (HIP kernels); no CPU fallback exists")
    return torch.device("cuda")


def _labels(labels):
    if isinstance(labels, torch.Tensor):
        t = labels
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(labels, dtype=np.float32)))
    return N.f32c(t.to(_dev()))


class Metric:
    """F1 / penalised DCG of the top-k prefix, averaged over the batch (float64 on device)."""

    @classmethod
    def _run(cls, labels, k_s, penalty=-1):
        y = _labels(labels)
        k = torch.as_tensor(np.asarray(k_s), dtype=torch.int32).to(y.device).contiguous()
        _, f1, dcg, sums = ops.cut_metrics(None, y, k_in=k, penalty=penalty)
        return sums.cpu().numpy() / y.shape[0]

    @classmethod
    def f1(cls, labels, k_s):
        return float(cls._run(labels, k_s)[0])

    @classmethod
    def dcg(cls, labels, k_s, penalty=-1):
        return float(cls._run(labels, k_s, penalty)[1])

    @classmethod
    def _task(cls, labels, predictions):
        y = _labels(labels)
        p = torch.as_tensor(np.asarray(predictions), dtype=torch.float32).to(y.device).contiguous() \
            if not torch.is_tensor(predictions) else N.f32c(predictions.detach().to(y.device))
        B, S = y.shape
        dcg = torch.empty((B,), dtype=torch.float64, device=y.device)
        auc = torch.empty((B,), dtype=torch.float64, device=y.device)
        sums = torch.empty((3,), dtype=torch.float64, device=y.device)
        N.call("rlt_task_metrics", N.ptr(y), N.ptr(p.reshape(B, S)), B, S, N.ptr(dcg), N.ptr(auc), N.ptr(sums), N.stream())
        return sums.cpu().numpy(), B

    @classmethod
    def taskr_metric(cls, labels, predictions):
        """utils/metrics.py:40-57: mean over the batch of the DCG of each list re-ranked by the predictions."""
        sums, B = cls._task(labels, predictions)
        return float(sums[0] / B)

    @classmethod
    def taskc_metric(cls, labels, predictions):
        """utils/metrics.py:59-76: mean ROC AUC over the lists that hold both classes."""
        sums, _ = cls._task(labels, predictions)
        return float(sums[1] / sums[2])        # the reference divides by zero too when no list qualifies

    @classmethod
    def evaluate(cls, output, labels):
        """output: cut distribution (B,S,1) or (B,S) on the GPU.  Returns (k (B,) int32 tensor,
        mean F1, mean DCG as 0-d float64 tensors) without leaving the device (run.py:137-145)."""
        p = N.f32c(output.detach())
        y = N.f32c(labels)
        if p.dim() == 3 and p.shape[2] == 2:
            # BiCut (run.py:131-136): k = S when every position says continue, else the first truncate position + 1
            cont = p[..., 1] > p[..., 0]                       # argmax == 1 (ties -> class 0)
            S = cont.shape[1]
            first0 = torch.where(~cont, torch.arange(S, device=p.device).expand_as(cont), torch.full_like(cont, S, dtype=torch.long)).min(dim=1).values
            k_in = torch.where(first0 >= S, torch.full_like(first0, S), first0 + 1).to(torch.int32).contiguous()
            k, _f1, _dcg, sums = ops.cut_metrics(None, y, k_in=k_in)
        else:
            k, _f1, _dcg, sums = ops.cut_metrics(p, y)
        mean = sums / y.shape[0]
        return k, mean[0], mean[1]

    @classmethod
    def step(cls, criterion, output, labels):
        """criterion(output, labels) and evaluate(cut distribution, labels) of one training step (run.py:126,137-145).
        Criteria built on the reward kernel (ChoopyLoss / AttnCutLoss / DivLoss / MtCutLoss) produce the metrics in the
        same pass over p and the labels (`rlt_loss_metrics`); any other criterion is followed by `evaluate`.
        Returns (loss, k, mean F1, mean DCG)."""
        fused = getattr(criterion, "forward_with_metrics", None)
        if fused is not None:
            loss, k, sums = fused(output, labels)
            mean = sums / labels.shape[0]
            return loss, k, mean[0], mean[1]
        loss = criterion(output, labels)
        cut = output[-1] if isinstance(output, (list, tuple)) else output
        k, f1, dcg = cls.evaluate(cut, labels)
        return loss, k, f1, dcg


class Metric_for_Loss:
    """Reward of ONE (list, k) pair, as the reference's per-element interface (utils/metrics.py:85-101).
    The criteria in utils/losses.py do not call this - they build the whole reward matrix in one
    kernel - it exists for drop-in completeness (plots, notebooks)."""

    @classmethod
    def _reward(cls, label, k, metric, penalty=-1):
        y = _labels(label).reshape(1, -1)
        r = ops.reward_matrix(y, metric, penalty=penalty)
        return r[0, k - 1].clone()

    @classmethod
    def f1(cls, label, k: int):
        return cls._reward(label, k, N.METRIC_F1)

    @classmethod
    def dcg(cls, label, k: int, penalty: int = -1):
        return cls._reward(label, k, N.METRIC_DCG, penalty)
