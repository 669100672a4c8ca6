"""The three task heads shared by MtAttnCut / MtChoopy (models/MtAttnCut.py:11-19,24-29)."""
from rlt_hip import native as N
from rlt_hip import ops
from . import _common as C

_KIND = {"classi": N.HEAD_SIGMOID, "rerank": N.HEAD_IDENTITY, "decison_layer": N.HEAD_SOFTMAX}


def mt_heads(module, h, S, B):
    names = C.pick_tasks(module.num_tasks)
    ws, bs, kinds = [], [], []
    for name in names:
        holder = getattr(module, name)
        lin = holder if name == "rerank" else getattr(holder, "0")
        ws.append(lin.weight)
        bs.append(lin.bias)
        kinds.append(_KIND[name])
    return ops.heads(h, ws, bs, kinds, S, B)
