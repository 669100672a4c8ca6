"""Choopy on the HIP hot path - drop-in for the reference's models/Choopy.py:6-23."""
import torch
from torch import nn

from rlt_hip import native as N
from rlt_hip import ops
from . import _common as C


class Choopy(nn.Module):
    def __init__(self, seq_len: int = 300, d_model: int = 128, n_head: int = 8, num_layers: int = 3, dropout=0.2):
        super().__init__()
        self.seq_len, self.n_head, self.dropout = seq_len, n_head, dropout
        self.position_encoding = nn.Parameter(torch.randn(seq_len, 127), requires_grad=True)
        self.attention_layer = C.encoder_params(d_model, n_head, num_layers, dropout)
        self.decison_layer = C.head_params(d_model)

    def forward(self, x):
        x = C.check_input(x)
        drop_p = C.check_dropout(self, self.dropout)
        B, S, _ = x.shape
        if S != self.seq_len:
            raise ValueError(f"Choopy was built for seq_len={self.seq_len}, got {S}")
        h = ops.choopy_embed(x, self.position_encoding)
        h = C.encoder(h, self.attention_layer, self.n_head, S, B, drop_p)
        head = getattr(self.decison_layer, "0")
        return ops.heads(h, [head.weight], [head.bias], [N.HEAD_SOFTMAX], S, B)[0]
