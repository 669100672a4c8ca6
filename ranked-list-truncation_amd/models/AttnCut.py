"""AttnCut on the HIP hot path - drop-in for the reference's models/AttnCut.py:5-20."""
from torch import nn

from rlt_hip import native as N
from rlt_hip import ops
from . import _common as C


class AttnCut(nn.Module):
    """BiLSTM(2 layers, H=128) -> encoder layer(s) with list-axis attention -> Linear -> softmax over
    positions.  Same constructor, state_dict keys and output shape (B,S,1) as the reference."""

    def __init__(self, input_size: int = 3, d_model: int = 256, n_head: int = 4, num_layers: int = 1,
                 dropout: float = 0.4):
        super().__init__()
        self.n_head, self.dropout = n_head, dropout
        self.encoding_layer = C.bilstm_params(input_size)
        self.attention_layer = C.encoder_params(d_model, n_head, num_layers, dropout)
        self.decison_layer = C.head_params(d_model)

    def forward(self, x):
        x = C.check_input(x)
        drop_p = C.check_dropout(self, self.dropout)
        B, S, _ = x.shape
        h = C.bilstm(ops.to_position_major(x), self.encoding_layer, S, B)
        h = C.encoder(h, self.attention_layer, self.n_head, S, B, drop_p)
        head = getattr(self.decison_layer, "0")
        return ops.heads(h, [head.weight], [head.bias], [N.HEAD_SOFTMAX], S, B)[0]
