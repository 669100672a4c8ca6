"""Data-parallel training glue: flat parameter / gradient buckets, one RCCL all-reduce per step,
one fused Adam launch per step.

The reference is single-process (no torch.distributed anywhere).  Ranked lists
are independent apart from the list-axis attention and the batch-wide rerank hinge, which couple
the lists of ONE mini-batch; the multi-GPU semantics are therefore "shard-wise reference semantics"
(SURVEY.md section 8e): every rank runs the reference computation on its own sub-batch, losses are
averaged over ranks, i.e. gradients are averaged with ONE all-reduce of the flat fp32 bucket
(AttnCut: 1,846,785 params = 7.4 MB) over xGMI.  No collective sits on the data path.

`FlatModel` re-points every parameter (and its .grad) of an nn.Module at views of two flat fp32
buffers, so that zero_grad is one memset, the all-reduce is one collective on one buffer, and
`FusedAdam` (torch.optim.Adam semantics: coupled L2, bias correction; run.py:104) is one
`rlt_adam_step` launch.
"""
import os

import torch
import torch.distributed as dist

from . import native as N

# RLT_FORCE_DIST=1 (tests / one-GPU rehearsals): run every collective of the step even in a ONE-rank process group, so that
# the RCCL code path (all-reduce AVG on the flat bucket, parameter broadcast) executes on a one-GPU box.
FORCE_COLLECTIVES = os.environ.get("RLT_FORCE_DIST") == "1"


class FlatModel:
    def __init__(self, model: torch.nn.Module):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("model has no trainable parameters")
        dev = params[0].device
        # 4-float (16-byte) aligned slots so every view keeps the alignment the kernels want
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.numel = total
        self.params = params
        with torch.no_grad():
            for p, o in zip(params, offs):
                n = p.numel()
                self.flat_param[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[o:o + n].view(p.shape)
                p.grad = self.flat_grad[o:o + n].view(p.shape)

    def zero_grad(self):
        """One memset; keeps the .grad views alive (autograd then accumulates in place)."""
        self.flat_grad.zero_()
        for p in self.params:          # a None grad would make autograd allocate a fresh tensor
            if p.grad is None or p.grad.data_ptr() < self.flat_grad.data_ptr():
                raise RuntimeError("a parameter lost its flat gradient view; use FlatModel.zero_grad(), "
                                   "not optimizer.zero_grad(set_to_none=True)")

    def all_reduce_grads(self, group=None):
        """Average the flat gradient bucket over the ranks (one collective)."""
        if not dist.is_available() or not dist.is_initialized():
            return
        world = dist.get_world_size(group)
        if world == 1 and not FORCE_COLLECTIVES:
            return
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.AVG, group=group)     # RCCL over xGMI
        else:                                                                       # gloo (CPU tests, rehearsals)
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=group)
            self.flat_grad.mul_(1.0 / world)

    def broadcast_params(self, src=0, group=None):
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES):
            dist.broadcast(self.flat_param, src=src, group=group)


class FusedAdam:
    """torch.optim.Adam(lr, betas, eps, weight_decay) on a FlatModel, one HIP launch per step."""

    def __init__(self, flat: FlatModel, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.flat_param)
        self.exp_avg_sq = torch.zeros_like(flat.flat_param)
        self.steps = 0

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self):
        f = self.flat
        if not f.flat_param.is_cuda:
            raise RuntimeError("FusedAdam runs on the GPU (rlt_adam_step); no CPU fallback exists")
        self.steps += 1
        N.call("rlt_adam_step", N.ptr(f.flat_param), N.ptr(f.flat_grad), N.ptr(self.exp_avg), N.ptr(self.exp_avg_sq),
               f.numel, self.steps, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, N.stream())

    def state_dict(self):
        return {"steps": self.steps, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}


def shard_batch(x, y, rank, world):
    """Equal contiguous shards of a global batch (the lists of one shard attend to each other)."""
    n = x.shape[0]
    if n % world:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]


def shard_bounds(n, rank, world):
    """[lo, hi) of rank's contiguous shard of an n-list batch when n need not divide: the first n % world ranks take one
    list more; a rank may get an empty shard (n < world).  The shards of all ranks partition range(n) exactly."""
    per, extra = divmod(n, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)
