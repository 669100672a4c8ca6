"""BiCut on the HIP hot path - drop-in for the reference's models/Bicut.py:5-21:
BiLSTM (2 layers, 128 hidden) -> Linear(256, fc) -> ReLU -> Linear(fc, 2) -> Dropout -> softmax over the two classes
{0: truncate, 1: continue} at every position.  Same constructor, state_dict keys and (B,S,2) output."""
from torch import nn

from rlt_hip import ops
from . import _common as C


class BiCut(nn.Module):
    def __init__(self, input_size=231449, lstm_hiden_size=128, lstm_layers=2, fc_dimensions=256, dropout=0.4):
        super().__init__()
        if lstm_hiden_size != 128 or lstm_layers != 2:
            raise ValueError("the HIP BiLSTM kernel is specialised for 2 layers of hidden size 128 (the reference's defaults)")
        self.dropout = dropout
        self.bilstm = C.bilstm_params(input_size, lstm_hiden_size)
        self.fc = C.ParamTree(nn.Linear(in_features=lstm_hiden_size * 2, out_features=fc_dimensions))
        self.softmax = C.ParamTree(nn.Sequential(nn.ReLU(), nn.Linear(in_features=fc_dimensions, out_features=2),
                                                 nn.Dropout(dropout), nn.Softmax(dim=2)))

    def forward(self, x):
        x = C.check_input(x)
        drop_p = C.check_dropout(self, self.dropout)
        B, S, _ = x.shape
        h = C.bilstm(ops.to_position_major(x), self.bilstm, S, B)                         # (S*B, 256)
        h = ops.linear(h, self.fc.weight, self.fc.bias, relu=True)                        # fc + the Sequential's ReLU
        head = getattr(self.softmax, "1")
        z = ops.linear(h, head.weight, head.bias)                                         # (S*B, 2)
        return ops.pair_softmax(z, S, B, drop_p)                                          # (B, S, 2)
