"""PLECut on the HIP hot path - drop-in for the reference's models/PLECut.py:55-103: three experts; the class tower
mixes experts {0,1}, the rerank tower {1,2}, the cut tower all three."""
import torch
from torch import nn

from rlt_hip import ops
from . import _common as C
from .MMOECut import Expert, TowerClass, TowerCut, TowerRerank


class PLECut(nn.Module):
    def __init__(self, seq_len: int = 300, num_experts=3, input_size=3, encoding_size=128, d_model=256, n_head=2,
                 num_layers=1, dropout=0.1):
        super().__init__()
        if d_model != 2 * encoding_size:
            raise ValueError(f"d_model ({d_model}) must equal 2 * encoding_size ({2 * encoding_size})")
        if num_experts != 3:
            raise ValueError("PLECut's expert groups are fixed for 3 experts (models/PLECut.py:80-82)")
        self.seq_len, self.expert_hidden, self.n_head, self.dropout = seq_len, d_model, n_head, dropout
        self.pre_encoding = C.bilstm_params(input_size, encoding_size)
        self.experts = nn.ModuleList([Expert(d_model, n_head, num_layers, dropout) for _ in range(num_experts)])
        self.w_gates = nn.ParameterList(
            [nn.Parameter(torch.randn(encoding_size * seq_len * 2, n), requires_grad=True) for n in (2, 2, 3)])
        self.towers = nn.ModuleList([TowerClass(d_model), TowerRerank(d_model), TowerCut(d_model)])

    def forward(self, x):
        x = C.check_input(x)
        drop_p = C.check_dropout(self, self.dropout)
        B, S, _ = x.shape
        if S != self.seq_len:
            raise ValueError(f"PLECut was built for seq_len={self.seq_len}, got {S}")
        h = C.bilstm(ops.to_position_major(x), self.pre_encoding, S, B)
        eo = [C.encoder(h, e.attention_layer, self.n_head, S, B, drop_p) for e in self.experts]
        groups = [eo[:2], eo[1:], eo]                                                     # models/PLECut.py:80-82
        outs = []
        for w_gate, grp, tower in zip(self.w_gates, groups, self.towers):
            gate = ops.MMOEGateFn.apply(h, S, B, w_gate)                                  # (1,B,len(grp))
            mixed = ops.MMOEMixFn.apply(gate, S, B, *grp)[0]
            lin = tower.linear
            outs.append(ops.heads(mixed, [lin.weight], [lin.bias], [tower.kind], S, B)[0])
        return outs
