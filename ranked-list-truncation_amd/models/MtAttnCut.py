"""MtAttnCut on the HIP hot path - drop-in for the reference's models/MtAttnCut.py:4-29."""
from torch import nn

from rlt_hip import ops
from . import _common as C
from ._mt import mt_heads


class MtAttnCut(nn.Module):
    def __init__(self, input_size: int = 3, d_model: int = 256, n_head: int = 4, num_layers: int = 1,
                 num_tasks: float = 3, dropout: float = 0.4):
        super().__init__()
        self.num_tasks, self.n_head, self.dropout = num_tasks, n_head, dropout
        self.pre_encoding = C.bilstm_params(input_size)
        self.encoding_layer = C.encoder_params(d_model, n_head, num_layers, dropout)
        self.classi = C.head_params(d_model)
        self.rerank = C.ParamTree(nn.Linear(d_model, 1))
        self.decison_layer = C.head_params(d_model)

    def forward(self, x):
        x = C.check_input(x)
        drop_p = C.check_dropout(self, self.dropout)
        B, S, _ = x.shape
        h = C.bilstm(ops.to_position_major(x), self.pre_encoding, S, B)
        h = C.encoder(h, self.encoding_layer, self.n_head, S, B, drop_p)
        return mt_heads(self, h, S, B)
