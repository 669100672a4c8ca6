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
    def _run(cls, labels, k_s):
        y = _labels(labels)
        k = torch.as_tensor(np.asarray(k_s), dtype=torch.int32).to(y.device).contiguous()
        _, f1, dcg, sums = ops.cut_metrics(None, y, k_in=k)
        return sums.cpu().numpy() / y.shape[0]

    @classmethod
    def f1(cls, labels, k_s):
        return float(cls._run(labels, k_s)[0])

    @classmethod
    def dcg(cls, labels, k_s, penalty=-1):
        if penalty != -1:
            raise NotImplementedError("only the reference's default penalty=-1 is implemented")
        return float(cls._run(labels, k_s)[1])

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


class Metric_for_Loss:
    """Reward of ONE (list, k) pair, as the reference's per-element interface (utils/metrics.py:85-101).
    The criteria in utils/losses.py do not call this - they build the whole reward matrix in one
    kernel - it exists for drop-in completeness (plots, notebooks)."""

    @classmethod
    def _reward(cls, label, k, metric):
        y = _labels(label).reshape(1, -1)
        r = ops.reward_matrix(y, metric)
        return r[0, k - 1].clone()

    @classmethod
    def f1(cls, label, k: int):
        return cls._reward(label, k, N.METRIC_F1)

    @classmethod
    def dcg(cls, label, k: int, penalty: int = -1):
        if penalty != -1:
            raise NotImplementedError("only the reference's default penalty=-1 is implemented")
        return cls._reward(label, k, N.METRIC_DCG)
