"""Shared building blocks of the HIP model classes.

Parameters are held in `ParamTree`s that mirror the module tree (and therefore the
`state_dict()` keys, shapes, order and init distributions) of the stock torch modules the
reference builds (nn.LSTM, nn.TransformerEncoder, nn.Sequential(nn.Linear, ...)); the forward
pass never calls those modules - it runs the HIP kernels through `rlt_hip.ops`.
"""
import torch
from torch import nn

from rlt_hip import native as N
from rlt_hip import ops


class ParamTree(nn.Module):
    """Parameter-only mirror of `src`: same names, shapes, order and (cloned) initial values."""

    def __init__(self, src: nn.Module):
        super().__init__()
        for name, p in src.named_parameters(recurse=False):
            self.register_parameter(name, nn.Parameter(p.detach().clone(), requires_grad=p.requires_grad))
        for name, child in src.named_children():
            sub = ParamTree(child)
            if next(sub.parameters(), None) is not None:
                self.add_module(name, sub)

    def forward(self, *a, **k):          # pragma: no cover - holders are never called
        raise RuntimeError("ParamTree only holds parameters")


def bilstm_params(input_size, hidden=128):
    # same construction as models/AttnCut.py:8
    return ParamTree(nn.LSTM(input_size=input_size, hidden_size=hidden, num_layers=2,
                             batch_first=True, bidirectional=True))


def encoder_params(d_model, n_head, num_layers, dropout):
    # same construction as models/AttnCut.py:9-10 (post-norm, ReLU, FF 2048, eps 1e-5)
    layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=n_head, dropout=dropout)
    return ParamTree(nn.TransformerEncoder(layer, num_layers=num_layers, enable_nested_tensor=False))


def head_params(d_model):
    # nn.Sequential(nn.Linear(d_model, 1), <activation>): parameters live under key "0"
    return ParamTree(nn.Sequential(nn.Linear(d_model, 1)))


def check_input(x, ndim=3):
    if not x.is_cuda:
        raise RuntimeError("the HIP models run on the GPU only: move the model and its inputs to 'cuda' "
                           "(no CPU fallback exists)")
    if x.dim() != ndim:
        raise ValueError(f"expected a {ndim}-d input, got {tuple(x.shape)}")
    return N.f32c(x)


def check_dropout(module, p):
    """Effective dropout probability: p in train(), 0 in eval() (nn.Dropout semantics)."""
    if not 0.0 <= p < 1.0:
        raise ValueError(f"dropout must be in [0, 1), got {p}")
    return float(p) if module.training else 0.0


def bilstm(x_pm, params, S, B):
    """2 stacked bidirectional layers, position-major in and out (one call into the library per direction of travel)."""
    return ops.bilstm(x_pm, params, S, B)


def encoder(x_pm, params, n_head, S, B, drop_p=0.0):
    """Stack of post-norm encoder layers with list-axis attention.  drop_p > 0 applies the four
    dropouts of nn.TransformerEncoderLayer: attention probabilities, dropout1 (attention branch),
    the FFN hidden dropout and dropout2 (FFN branch)."""
    h = x_pm
    for layer in params.layers.children():
        h = ops.encoder_layer(h, layer, S, B, n_head, drop_p)
    return h


def pick_tasks(num_tasks):
    """Head list for the float task code of models/MtAttnCut.py:27-29."""
    if num_tasks == 3:
        return ["classi", "rerank", "decison_layer"]
    if num_tasks == 2.1:
        return ["classi", "decison_layer"]
    return ["rerank", "decison_layer"]
