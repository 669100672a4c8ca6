"""MOECut on the HIP hot path - drop-in for the reference's models/MOECut.py:55-108 (MMOECut with ONE gate matrix
shared by all towers)."""
import torch
from torch import nn

from rlt_hip import ops
from . import _common as C
from .MMOECut import Expert, TowerClass, TowerCut, TowerRerank


class MOECut(nn.Module):
    def __init__(self, seq_len: int = 300, num_experts=3, num_tasks=3, input_size=3, encoding_size=128,
                 d_model=256, n_head=4, num_layers=1, dropout=0.2):
        super().__init__()
        if d_model != 2 * encoding_size:
            raise ValueError(f"d_model ({d_model}) must equal 2 * encoding_size ({2 * encoding_size})")
        self.seq_len, self.expert_hidden, self.n_head, self.dropout = seq_len, d_model, n_head, dropout
        self.pre_encoding = C.bilstm_params(input_size, encoding_size)
        self.experts = nn.ModuleList([Expert(d_model, n_head, num_layers, dropout) for _ in range(num_experts)])
        self.w_gates = nn.Parameter(torch.randn(encoding_size * seq_len * 2, num_experts), requires_grad=True)
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
            raise ValueError(f"MOECut was built for seq_len={self.seq_len}, got {S}")
        h = C.bilstm(ops.to_position_major(x), self.pre_encoding, S, B)                  # (S*B, 256)
        expert_out = [C.encoder(h, e.attention_layer, self.n_head, S, B, drop_p) for e in self.experts]
        gates = ops.MMOEGateFn.apply(h, S, B, self.w_gates)                               # (1,B,n_e)
        mixed = ops.MMOEMixFn.apply(gates, S, B, *expert_out)[0]                          # (S*B,E), shared by the towers
        lins = [t.linear for t in self.towers]
        return list(ops.heads(mixed, [l.weight for l in lins], [l.bias for l in lins], [t.kind for t in self.towers], S, B))
