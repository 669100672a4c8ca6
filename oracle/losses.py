"""CPU restatement of the reward losses (TEST INFRASTRUCTURE ONLY).

Rows L1..L8 of SURVEY.md section 8(a):

    reward_f1 / reward_dcg     utils/metrics.py:85-91, :93-101  (Metric_for_Loss)
    reward_matrix              utils/losses.py:57-65 = :81-89 = :217-225 (the B*S double loop)
    ChoopyLoss                 utils/losses.py:48-68
    AttnCutLoss                utils/losses.py:71-96
    DivLoss                    utils/losses.py:194-233
    RerankLoss                 utils/losses.py:99-141
    MtCutLoss                  utils/losses.py:164-191

Two reward-matrix builders are provided and tested against each other and against the
golden fixtures:

  * `reward_matrix_loop`  - loop-faithful: one scalar fp32 evaluation per (list, k), in the
    reference's operation order.  O(B*S^2); this is what the reference actually costs and
    what `bench.py` times as the CPU baseline's loss leg.
  * `reward_matrix`       - closed form on prefix sums (what the HIP kernel implements):
    F1@k = 2c_k/(k+N) evaluated in the reference's fp32 operation order, DCG@k = prefix sum
    of (+-1)/log2(j+2).
"""
import torch
from torch import nn

from .metrics import dcg_coef


# ----------------------------------------------------------------------------- rewards
def reward_f1(label: torch.Tensor, k: int) -> torch.Tensor:
    """F1 of the first k entries of one 0/1 label vector (fp32 scalar tensor)."""
    n_rel = label.sum()
    hit = label[:k].sum()
    prec = hit / k
    rec = hit / n_rel if n_rel != 0 else torch.tensor(0)
    total = prec + rec
    return (prec * rec * 2) / total if total != 0 else torch.tensor(0)


def reward_dcg(label: torch.Tensor, k: int, penalty: int = -1) -> torch.Tensor:
    """Penalised DCG of the first k entries (relevant: +1/log2(j+2), other: penalty/log2(j+2))."""
    head = label[:k]
    coef = torch.tensor(dcg_coef(label.shape[0])[:k])          # float64 list -> fp32 tensor
    rel = (head == 1.).float()
    irr = (head != 1.).float()
    return (rel / coef + (irr / coef) * penalty).sum()


def reward_matrix_loop(labels: torch.Tensor, metric: str) -> torch.Tensor:
    fn = reward_f1 if metric == 'f1' else reward_dcg
    r = torch.ones(labels.shape, dtype=torch.float32)
    for i in range(labels.shape[0]):
        for j in range(labels.shape[1]):
            r[i][j] = fn(labels[i], j + 1)
    return r


def reward_matrix(labels: torch.Tensor, metric: str, penalty: float = -1) -> torch.Tensor:
    labels = labels.float()
    n_pos = labels.shape[1]
    if metric == 'f1':
        hits = labels.cumsum(dim=1)
        ks = torch.arange(1, n_pos + 1, dtype=torch.float32)
        n_rel = hits[:, -1:]
        prec = hits / ks
        rec = torch.where(n_rel != 0, hits / torch.where(n_rel != 0, n_rel, torch.ones_like(n_rel)),
                          torch.zeros_like(hits))
        total = prec + rec
        safe = torch.where(total != 0, total, torch.ones_like(total))
        return torch.where(total != 0, (prec * rec * 2) / safe, torch.zeros_like(total))
    coef = torch.tensor(dcg_coef(n_pos))
    gain = (labels == 1.).float() / coef + ((labels != 1.).float() / coef) * penalty
    return gain.cumsum(dim=1)


def reward_distribution(r: torch.Tensor, tau: float) -> torch.Tensor:
    # utils/losses.py:226-228: exp(r/tau) normalised over positions, no max-subtraction
    q = torch.exp(r / tau)
    return q / q.sum(dim=1, keepdim=True)


def _rewards(labels, metric, loop):
    return reward_matrix_loop(labels, metric) if loop else reward_matrix(labels, metric)


# ------------------------------------------------------------------------------ criteria
class ChoopyLoss(nn.Module):
    """Negative expected reward under the predicted cut distribution."""

    def __init__(self, metric: str = 'f1', loop: bool = False):
        super().__init__()
        self.metric, self.loop = metric, loop

    def forward(self, output, labels):
        r = _rewards(labels, self.metric, self.loop)
        return -(output.squeeze(2) * r).sum() / output.shape[0]


class AttnCutLoss(nn.Module):
    """Cross entropy between softmax(r/tau) and the predicted cut distribution."""

    def __init__(self, metric: str = 'f1', tau: float = 0.95, loop: bool = False):
        super().__init__()
        self.metric, self.tau, self.loop = metric, tau, loop

    def forward(self, output, labels):
        q = reward_distribution(_rewards(labels, self.metric, self.loop), self.tau)
        return -(torch.log(output.squeeze(2)) * q).sum() / output.shape[0]


class DivLoss(nn.Module):
    """KL or Jensen-Shannon divergence between softmax(r/tau) and the predicted distribution."""

    def __init__(self, metric: str = 'f1', tau: float = 0.85, div_type: str = 'kl',
                 augmented: bool = True, loop: bool = False):
        super().__init__()
        self.metric, self.div_type, self.loop = metric, div_type, loop
        self.tau = tau if augmented else 1.
        self.kl = nn.KLDivLoss(reduction='batchmean')        # sum(t*(log t - input)) / B

    def forward(self, output, labels):
        p = output.squeeze(2)
        q = reward_distribution(_rewards(labels, self.metric, self.loop), self.tau)
        if self.div_type == 'kl':
            return self.kl(p.log(), q)
        log_mid = ((p + q) / 2).log()
        return (self.kl(log_mid, q) + self.kl(log_mid, p)) / 2


class RerankLoss(nn.Module):
    """Batch-wide hinge between the mean score of irrelevant and of relevant documents."""

    def __init__(self, margin: float = 5e-4, reduction: str = 'mean'):
        super().__init__()
        self.margin, self.reduction = margin, reduction

    def forward(self, output, labels):
        rel = labels == 1.
        irr = labels == 0.
        n_rel, n_irr = rel.sum().item(), irr.sum().item()
        if n_rel == 0 or n_irr == 0:
            return torch.tensor(0., requires_grad=True)
        score = output.squeeze(2)
        gap = (irr * score).sum() / n_irr - (rel * score).sum() / n_rel + self.margin
        # python max(tensor0, gap): gap only when strictly positive (utils/losses.py:141)
        return gap if gap > 0 else torch.tensor(0., requires_grad=True)


class MtCutLoss(nn.Module):
    """cut JS-divergence + w_r * rerank hinge + w_c * BCE(classifier)."""

    def __init__(self, metric: str = 'f1', rerank_weight: float = 0.5, classi_weight: float = 0.5,
                 num_tasks: float = 3, loop: bool = False):
        super().__init__()
        self.rerank_weight, self.classi_weight, self.num_tasks = rerank_weight, classi_weight, num_tasks
        self.cutloss = DivLoss(metric=metric, div_type='js', augmented=True, loop=loop)
        self.rerankloss = RerankLoss()
        self.classiloss = nn.BCELoss()                          # mean over B*S, log clamped at -100

    def forward(self, output, labels):
        y_class = y_rerank = None
        if self.num_tasks == 3:
            y_class, y_rerank, y_cut = output
        elif self.num_tasks == 2.1:
            y_class, y_cut = output
        else:
            y_rerank, y_cut = output
        loss = self.cutloss(y_cut, labels)
        if y_rerank is not None:
            loss = loss + self.rerankloss(y_rerank, labels) * self.rerank_weight
        if y_class is not None:
            loss = loss + self.classiloss(y_class.squeeze(2), labels) * self.classi_weight
        return loss


def bicut_last_truncate(output: torch.Tensor) -> torch.Tensor:
    """utils/losses.py:21-30 (BiCutLoss.slice_index) for a batch: index of the LAST position whose argmax over the
    two classes is 0 (truncate), or S when every position says 1 (continue).  argmax ties go to class 0."""
    temp = (output[..., 1] > output[..., 0])                     # True <=> argmax == 1
    S = temp.shape[1]
    pos = torch.arange(S).expand_as(temp)
    last0 = torch.where(~temp, pos, torch.full_like(pos, -1)).max(dim=1).values
    return torch.where(last0 < 0, torch.full_like(last0, S), last0)


class BiCutLoss(nn.Module):
    """utils/losses.py:11-45.  mask = positions up to and including the last truncate decision; r from the labels:
    'nci': label 1 -> (0, -1/log2(j+2)), label 0 -> (0, (j+1)/alpha); any other metric: label 1 -> ((1-alpha)/r, 0),
    label 0 -> (0, alpha/(1-r)); loss = sum(output * mask * r) / B."""

    def __init__(self, alpha: float = 0.65, r: float = 0.0971134020, metric: str = 'nci'):
        super().__init__()
        self.metric, self.alpha, self.r = metric, alpha, r

    def reward(self, labels: torch.Tensor) -> torch.Tensor:
        B, S = labels.shape
        pos = labels == 1
        out = torch.zeros(B, S, 2, dtype=torch.float32)
        j = torch.arange(S, dtype=torch.float64)
        if self.metric == 'nci':
            out[..., 1] = torch.where(pos, (-1.0 / torch.log2(j + 2)).float().expand(B, S),
                                      ((j + 1) / self.alpha).float().expand(B, S))
        else:
            out[..., 0] = torch.where(pos, torch.tensor((1 - self.alpha) / self.r, dtype=torch.float32), torch.tensor(0.0))
            out[..., 1] = torch.where(pos, torch.tensor(0.0), torch.tensor(self.alpha / (1 - self.r), dtype=torch.float32))
        return out

    def forward(self, output: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        idx = bicut_last_truncate(output.detach())
        S = output.shape[1]
        mask = (torch.arange(S).unsqueeze(0) <= idx.unsqueeze(1)).to(output.dtype).unsqueeze(2)
        return (output * mask * self.reward(labels)).sum() / output.shape[0]


class WassDistLoss(nn.Module):
    """utils/losses.py:236-311: entropic optimal transport between the B predicted cut
    distributions and the B label vectors of a batch.  Cost C_ij = sum_s (p_is - y_js)^2, uniform marginals 1/B,
    log-domain Sinkhorn with regularisation eps: per iteration
        u <- u + eps * (log(1/B + 1e-8) - logsumexp_j((-C_ij + u_i + v_j) / eps))
        v <- v + eps * (log(1/B + 1e-8) - logsumexp_i((-C_ij + u_i + v_j) / eps))       (with the new u)
    at most max_iter times, stopping after the iteration in which sum_i |u_i - u_i(previous)| < 0.1;
    loss = sum_ij exp((-C_ij + u_i + v_j) / eps) * C_ij, differentiated through the iterations."""

    def __init__(self, eps: float = 1e-3, max_iter: int = 100, metric: str = 'f1', tau: float = 0.95, reduction='mean'):
        super().__init__()
        self.eps, self.max_iter, self.reduction = eps, max_iter, reduction

    def forward(self, output: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        p = output.squeeze()
        C = ((p.unsqueeze(-2) - labels.unsqueeze(-3)).abs() ** 2).sum(-1)             # (B,B)
        B = p.shape[-2]
        log_m = torch.log(torch.full((B,), 1.0 / B) + 1e-8)
        u, v = torch.zeros(B), torch.zeros(B)

        def modified(u_, v_):
            return (-C + u_.unsqueeze(-1) + v_.unsqueeze(-2)) / self.eps

        for _ in range(self.max_iter):
            u_prev = u
            u = (log_m - torch.logsumexp(modified(u, v), dim=-1)).mul(self.eps).add(u)
            v = (log_m - torch.logsumexp(modified(u, v).transpose(-2, -1), dim=-1)).mul(self.eps).add(v)
            if (u - u_prev).abs().sum(-1).mean().item() < 1e-1:
                break
        cost = (torch.exp(modified(u, v)) * C).sum(dim=(-2, -1))
        return cost.mean() if self.reduction == 'mean' else cost.sum()
