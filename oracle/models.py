"""CPU restatement of the reference truncation models (TEST INFRASTRUCTURE ONLY).

Same constructor signatures, same `state_dict()` keys and shapes, same arithmetic as the
reference classes, built from stock torch-CPU modules.  Reference lines followed:

    AttnCut    models/AttnCut.py:5-20
    Choopy     models/Choopy.py:6-23
    MtAttnCut  models/MtAttnCut.py:4-29
    MtChoopy   models/MtChoopy.py:5-32
    MMOECut    models/MMOECut.py:56-110  (Expert :6-14, TowerCut :17-27,
                                          TowerClass :30-40, TowerRerank :43-53)
    MOECut     models/MOECut.py:55-108   (one shared gate)
    PLECut     models/PLECut.py:55-103   (3 experts; gates over experts {0,1}, {1,2}, {0,1,2})
    BiCut      models/Bicut.py:5-21      (BiLSTM -> Linear(256,256) -> ReLU -> Linear(256,2) -> Dropout -> softmax over
                                          the two classes {0: truncate, 1: continue} at every position)

The one behaviour that is easy to miss: every encoder layer is a
`nn.TransformerEncoderLayer` with `batch_first=False` that is fed a (B, S, E) tensor, so
self-attention runs over axis 0 - the B lists of the mini-batch - independently at each
of the S positions.  `_encoder()` keeps exactly that construction.
"""
import torch
from torch import nn


def _encoder(d_model, n_head, num_layers, dropout):
    # post-norm, ReLU, dim_feedforward=2048, eps=1e-5, batch_first=False (torch defaults),
    # as in models/AttnCut.py:9-10.
    layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=n_head, dropout=dropout)
    return nn.TransformerEncoder(layer, num_layers=num_layers, enable_nested_tensor=False)


def _bilstm(input_size, hidden):
    # models/AttnCut.py:8 - 2 stacked bidirectional layers, batch_first, no dropout.
    return nn.LSTM(input_size=input_size, hidden_size=hidden, num_layers=2,
                   batch_first=True, bidirectional=True)


def _softmax_head(d_model):
    # Linear(E,1) then softmax over dim=1 (the S positions of one list), models/AttnCut.py:11-14
    return nn.Sequential(nn.Linear(d_model, 1), nn.Softmax(dim=1))


def _sigmoid_head(d_model):
    return nn.Sequential(nn.Linear(d_model, 1), nn.Sigmoid())


def _pick_tasks(num_tasks, y_class, y_rerank, y_cut):
    # models/MtAttnCut.py:27-29 - float task code: 3, 2.1 (class+cut), else (rerank+cut)
    if num_tasks == 3:
        return [y_class, y_rerank, y_cut]
    if num_tasks == 2.1:
        return [y_class, y_cut]
    return [y_rerank, y_cut]


class AttnCut(nn.Module):
    def __init__(self, input_size: int = 3, d_model: int = 256, n_head: int = 4,
                 num_layers: int = 1, dropout: float = 0.4):
        super().__init__()
        self.encoding_layer = _bilstm(input_size, 128)
        self.attention_layer = _encoder(d_model, n_head, num_layers, dropout)
        self.decison_layer = _softmax_head(d_model)      # (sic) reference attribute name

    def forward(self, x):
        h = self.encoding_layer(x)[0]
        h = self.attention_layer(h)
        return self.decison_layer(h)


class Choopy(nn.Module):
    def __init__(self, seq_len: int = 300, d_model: int = 128, n_head: int = 8,
                 num_layers: int = 3, dropout=0.2):
        super().__init__()
        self.seq_len = seq_len
        self.position_encoding = nn.Parameter(torch.randn(seq_len, 127), requires_grad=True)
        self.attention_layer = _encoder(d_model, n_head, num_layers, dropout)
        self.decison_layer = _softmax_head(d_model)

    def forward(self, x):
        # models/Choopy.py:19-20: score column followed by the 127 learned columns
        pe = self.position_encoding.expand(x.shape[0], self.seq_len, 127)
        h = torch.cat((x, pe), dim=2)
        h = self.attention_layer(h)
        return self.decison_layer(h)


class MtAttnCut(nn.Module):
    def __init__(self, input_size: int = 3, d_model: int = 256, n_head: int = 4,
                 num_layers: int = 1, num_tasks: float = 3, dropout: float = 0.4):
        super().__init__()
        self.num_tasks = num_tasks
        self.pre_encoding = _bilstm(input_size, 128)
        self.encoding_layer = _encoder(d_model, n_head, num_layers, dropout)
        self.classi = _sigmoid_head(d_model)
        self.rerank = nn.Linear(d_model, 1)
        self.decison_layer = _softmax_head(d_model)

    def forward(self, x):
        h = self.pre_encoding(x)[0]
        h = self.encoding_layer(h)
        return _pick_tasks(self.num_tasks, self.classi(h), self.rerank(h), self.decison_layer(h))


class MtChoopy(nn.Module):
    def __init__(self, seq_len: int = 300, d_model: int = 128, n_head: int = 8,
                 num_layers: int = 3, num_tasks: float = 3, dropout: float = 0.4):
        super().__init__()
        self.seq_len = seq_len
        self.num_tasks = num_tasks
        self.position_encoding = nn.Parameter(torch.randn(seq_len, 127), requires_grad=True)
        self.encoding_layer = _encoder(d_model, n_head, num_layers, dropout)
        self.classi = _sigmoid_head(d_model)
        self.rerank = nn.Linear(d_model, 1)
        self.decison_layer = _softmax_head(d_model)

    def forward(self, x):
        pe = self.position_encoding.expand(x.shape[0], self.seq_len, 127)
        h = torch.cat((x, pe), dim=2)
        h = self.encoding_layer(h)
        return _pick_tasks(self.num_tasks, self.classi(h), self.rerank(h), self.decison_layer(h))


class Expert(nn.Module):
    def __init__(self, d_model, n_head, num_layers, dropout: float = 0.2):
        super().__init__()
        self.attention_layer = _encoder(d_model, n_head, num_layers, dropout)

    def forward(self, x):
        return self.attention_layer(x)


class TowerCut(nn.Module):
    def __init__(self, d_model):
        super().__init__()
        self.cut_layer = _softmax_head(d_model)

    def forward(self, x):
        return self.cut_layer(x)


class TowerClass(nn.Module):
    def __init__(self, d_model):
        super().__init__()
        self.classification_layer = _sigmoid_head(d_model)

    def forward(self, x):
        return self.classification_layer(x)


class TowerRerank(nn.Module):
    # NB: the MMOE rerank tower ends in a softmax over positions (models/MMOECut.py:46-49),
    # unlike the raw `rerank` Linear of the Mt* models.
    def __init__(self, d_model):
        super().__init__()
        self.rerank_layer = _softmax_head(d_model)

    def forward(self, x):
        return self.rerank_layer(x)


class MMOECut(nn.Module):
    def __init__(self, seq_len: int = 300, num_experts=3, num_tasks=3, input_size=3,
                 encoding_size=128, d_model=256, n_head=4, num_layers=1, dropout=0.2):
        super().__init__()
        self.seq_len = seq_len
        self.expert_hidden = d_model
        self.pre_encoding = _bilstm(input_size, encoding_size)
        self.softmax = nn.Softmax(dim=1)
        self.experts = nn.ModuleList(
            [Expert(d_model, n_head, num_layers, dropout) for _ in range(num_experts)])
        # one (S*2H, n_e) gate matrix per task, N(0,1) init (models/MMOECut.py:68)
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
        h = self.pre_encoding(x)[0]                                   # (B,S,2H)
        expert_out = torch.stack([e(h) for e in self.experts])       # (n_e,B,S,E)
        flat = h.reshape(h.shape[0], -1)                              # (B,S*2H)
        outs = []
        for w_gate, tower in zip(self.w_gates, self.towers):
            gate = self.softmax(flat @ w_gate)                        # (B,n_e), softmax over experts
            mixed = (gate.t()[:, :, None, None] * expert_out).sum(dim=0)   # models/MMOECut.py:101-102
            outs.append(tower(mixed))
        return outs


class MOECut(nn.Module):
    """models/MOECut.py:55-108: MMOECut with ONE gate matrix shared by all towers."""

    def __init__(self, seq_len: int = 300, num_experts=3, num_tasks=3, input_size=3,
                 encoding_size=128, d_model=256, n_head=4, num_layers=1, dropout=0.2):
        super().__init__()
        self.seq_len = seq_len
        self.expert_hidden = d_model
        self.pre_encoding = _bilstm(input_size, encoding_size)
        self.softmax = nn.Softmax(dim=1)
        self.experts = nn.ModuleList(
            [Expert(d_model, n_head, num_layers, dropout) for _ in range(num_experts)])
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
        h = self.pre_encoding(x)[0]
        expert_out = torch.stack([e(h) for e in self.experts])
        gate = self.softmax(h.reshape(h.shape[0], -1) @ self.w_gates)              # models/MOECut.py:92
        mixed = (gate.t()[:, :, None, None] * expert_out).sum(dim=0)               # :98-99
        return [tower(mixed) for tower in self.towers]


class PLECut(nn.Module):
    """models/PLECut.py:55-103: three experts; the class tower mixes experts {0,1}, the rerank tower {1,2}, the cut
    tower all three (gate matrices of 2, 2 and 3 columns)."""

    def __init__(self, seq_len: int = 300, num_experts=3, input_size=3, encoding_size=128,
                 d_model=256, n_head=2, num_layers=1, dropout=0.1):
        super().__init__()
        self.seq_len = seq_len
        self.expert_hidden = d_model
        self.pre_encoding = _bilstm(input_size, encoding_size)
        self.softmax = nn.Softmax(dim=1)
        self.experts = nn.ModuleList(
            [Expert(d_model, n_head, num_layers, dropout) for _ in range(num_experts)])
        self.w_gates = nn.ParameterList(
            [nn.Parameter(torch.randn(encoding_size * seq_len * 2, n), requires_grad=True) for n in (2, 2, 3)])
        self.towers = nn.ModuleList([TowerClass(d_model), TowerRerank(d_model), TowerCut(d_model)])

    def forward(self, x):
        h = self.pre_encoding(x)[0]
        eo = [e(h) for e in self.experts]
        groups = [torch.stack(eo[:2]), torch.stack(eo[1:]), torch.stack(eo)]        # models/PLECut.py:80-82
        flat = h.reshape(h.shape[0], -1)
        outs = []
        for w_gate, grp, tower in zip(self.w_gates, groups, self.towers):
            gate = self.softmax(flat @ w_gate)
            outs.append(tower((gate.t()[:, :, None, None] * grp).sum(dim=0)))
        return outs


class BiCut(nn.Module):
    def __init__(self, input_size=231449, lstm_hiden_size=128, lstm_layers=2, fc_dimensions=256, dropout=0.4):
        super().__init__()
        self.bilstm = nn.LSTM(input_size=input_size, hidden_size=lstm_hiden_size, num_layers=lstm_layers,
                              batch_first=True, bidirectional=True)
        self.fc = nn.Linear(in_features=lstm_hiden_size * 2, out_features=fc_dimensions)
        self.softmax = nn.Sequential(nn.ReLU(), nn.Linear(in_features=fc_dimensions, out_features=2),
                                     nn.Dropout(dropout), nn.Softmax(dim=2))

    def forward(self, x):
        return self.softmax(self.fc(self.bilstm(x)[0]))          # (B,S,2)


MODEL_TABLE = {
    "attncut": AttnCut, "choopy": Choopy, "mtattncut": MtAttnCut,
    "mtchoopy": MtChoopy, "mmoecut": MMOECut, "moecut": MOECut, "plecut": PLECut, "bicut": BiCut,
}
