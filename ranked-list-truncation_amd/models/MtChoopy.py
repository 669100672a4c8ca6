"""MtChoopy on the HIP hot path - drop-in for the reference's models/MtChoopy.py:5-32."""
import torch
from torch import nn

from rlt_hip import ops
from . import _common as C
from ._mt import mt_heads


class MtChoopy(nn.Module):
    def __init__(self, seq_len: int = 300, d_model: int = 128, n_head: int = 8, num_layers: int = 3,
                 num_tasks: float = 3, dropout: float = 0.4):
        super().__init__()
        self.seq_len, self.num_tasks, self.n_head, self.dropout = seq_len, num_tasks, n_head, dropout
        self.position_encoding = nn.Parameter(torch.randn(seq_len, 127), requires_grad=True)
        self.encoding_layer = C.encoder_params(d_model, n_head, num_layers, dropout)
        self.classi = C.head_params(d_model)
        self.rerank = C.ParamTree(nn.Linear(d_model, 1))
        self.decison_layer = C.head_params(d_model)

    def forward(self, x):
        x = C.check_input(x)
        drop_p = C.check_dropout(self, self.dropout)
        B, S, _ = x.shape
        if S != self.seq_len:
            raise ValueError(f"MtChoopy was built for seq_len={self.seq_len}, got {S}")
        h = ops.choopy_embed(x, self.position_encoding)
        h = C.encoder(h, self.encoding_layer, self.n_head, S, B, drop_p)
        return mt_heads(self, h, S, B)
