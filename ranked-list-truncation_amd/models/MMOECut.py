"""MMOECut on the HIP hot path - drop-in for the reference's models/MMOECut.py:56-110."""
import torch
from torch import nn

from rlt_hip import native as N
from rlt_hip import ops
from . import _common as C


class Expert(nn.Module):
    """One encoder stack (models/MMOECut.py:6-14); holds parameters, run by MMOECut.forward."""

    def __init__(self, d_model, n_head, num_layers, dropout: float = 0.2):
        super().__init__()
        self.attention_layer = C.encoder_params(d_model, n_head, num_layers, dropout)


class _Tower(nn.Module):
    def __init__(self, d_model, attr):
        super().__init__()
        setattr(self, attr, C.head_params(d_model))
        self._attr = attr

    @property
    def linear(self):
        return getattr(getattr(self, self._attr), "0")


class TowerCut(_Tower):
    kind = N.HEAD_SOFTMAX

    def __init__(self, d_model):
        super().__init__(d_model, "cut_layer")


class TowerClass(_Tower):
    kind = N.HEAD_SIGMOID

    def __init__(self, d_model):
        super().__init__(d_model, "classification_layer")


class TowerRerank(_Tower):
    kind = N.HEAD_SOFTMAX          # the MMOE rerank tower ends in a softmax over positions (MMOECut.py:46-49)

    def __init__(self, d_model):
        super().__init__(d_model, "rerank_layer")


class MMOECut(nn.Module):
    def __init__(self, seq_len: int = 300, num_experts=3, num_tasks=3, input_size=3, encoding_size=128,
                 d_model=256, n_head=4, num_layers=1, dropout=0.2):
        super().__init__()
        if d_model != 2 * encoding_size:
            # the reference fails later, inside the first expert (models/MMOECut.py:88-90: the experts take the BiLSTM's
            # 2*encoding_size outputs as d_model inputs); say so at construction
            raise ValueError(f"d_model ({d_model}) must equal 2 * encoding_size ({2 * encoding_size})")
        self.seq_len, self.expert_hidden, self.n_head, self.dropout = seq_len, d_model, n_head, dropout
        self.pre_encoding = C.bilstm_params(input_size, encoding_size)
        self.experts = nn.ModuleList([Expert(d_model, n_head, num_layers, dropout) for _ in range(num_experts)])
        self.w_gates = nn.ParameterList(
            [nn.Parameter(torch.randn(encoding_size * seq_len * 2, num_experts), requires_grad=True)
             for _ in range(int(num_tasks))])
        if num_tasks == 3:
            towers = [TowerClass(d_model), TowerRerank(d_model), TowerCut(d_model)]
        elif num_tasks == 2.1:
            towers = [TowerClass(d_model), TowerCut(d_model)]
        elif num_tasks == 2.2:
            towers = [TowerRerank(d_model), TowerCut(d_model)]
        else:
            raise ValueError("num_tasks must be 3, 2.1 or 2.2")
        self.towers = nn.ModuleList(towers)

    def forward(self, x):
        x = C.check_input(x)
        drop_p = C.check_dropout(self, self.dropout)
        B, S, _ = x.shape
        if S != self.seq_len:
            raise ValueError(f"MMOECut was built for seq_len={self.seq_len}, got {S}")
        h = C.bilstm(ops.to_position_major(x), self.pre_encoding, S, B)                  # (S*B, 256)
        expert_out = [C.encoder(h, e.attention_layer, self.n_head, S, B, drop_p) for e in self.experts]
        gates = ops.MMOEGateFn.apply(h, S, B, *self.w_gates)                              # (n_tasks,B,n_e)
        mixed = ops.MMOEMixFn.apply(gates, S, B, *expert_out)                             # (n_tasks,S*B,E)
        outs = []
        for t, tower in enumerate(self.towers):
            lin = tower.linear
            outs.append(ops.heads(mixed[t], [lin.weight], [lin.bias], [tower.kind], S, B)[0])
        return outs
