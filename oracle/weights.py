"""Deterministic weight and input recipes shared by the golden-vector generator and the tests.

Build-owned (nothing here comes from the reference): fixtures therefore never need to
store weights - every side regenerates them from (state_dict key order, seed).
Uses numpy's frozen legacy `RandomState` stream so values do not depend on the torch or
numpy version.
"""
import math

import numpy as np
import torch


def fill_state_dict(model: torch.nn.Module, seed: int, gate_scale: float = None, pe_scale: float = None) -> None:
    """Overwrite every entry of `model.state_dict()` in key order.

    * LayerNorm weights ~ 1 + 0.1*U(-1,1); all biases ~ 0.1*U(-1,1)
    * `position_encoding` ~ N(0,1) (as the reference initialises it) * pe_scale (default 1; a small pe_scale lets the
      score column, the only list-dependent input of Choopy, decide the cut position: cases with many distinct k)
    * `w_gates.*` ~ N(0,1)*gate_scale (reference: gate_scale=1 => near one-hot gates;
      tests also use 1/sqrt(fan_in) so that the gate softmax is exercised away from saturation)
    * every other matrix ~ U(-1,1)/sqrt(fan_in)
    """
    rs = np.random.RandomState(seed)
    new = {}
    for key, ten in model.state_dict().items():
        shape = tuple(ten.shape)
        if key.endswith("position_encoding"):
            val = rs.standard_normal(shape) * (1.0 if pe_scale is None else pe_scale)
        elif ".w_gates." in "." + key or key.startswith("w_gates.") or key == "w_gates":
            scale = gate_scale if gate_scale is not None else 1.0 / math.sqrt(shape[0])
            val = rs.standard_normal(shape) * scale
        elif "norm" in key and key.endswith("weight"):
            val = 1.0 + 0.1 * rs.uniform(-1, 1, shape)
        elif "bias" in key:
            val = 0.1 * rs.uniform(-1, 1, shape)
        else:
            fan_in = shape[-1] if len(shape) > 1 else shape[0]
            val = rs.uniform(-1, 1, shape) / math.sqrt(fan_in)
        new[key] = torch.tensor(val, dtype=torch.float32)
    model.load_state_dict(new)


def synthetic_lists(batch: int, seq_len: int, n_features: int, seed: int, noise: float = 0.0):
    """robust04-shaped synthetic ranked lists.

    scores: per list, descending sort of N(3, 2.5^2); extra feature columns U(0,1);
    labels: Bernoulli(0.55*exp(-j/45)+0.02) at rank j, at least one positive per list.
    noise > 0 adds N(0, noise^2) to every score AFTER the sort (drawn last, so noise = 0 reproduces the plain lists):
    unsorted, list-specific inputs for the cases that want many distinct cut positions.
    Returns X (B,S,F) float32, y (B,S) float32 in {0,1}.
    """
    rs = np.random.RandomState(seed)
    scores = np.sort(rs.standard_normal((batch, seq_len)) * 2.5 + 3.0, axis=1)[:, ::-1]
    cols = [scores[:, :, None]]
    if n_features > 1:
        cols.append(rs.uniform(0, 1, (batch, seq_len, n_features - 1)))
    x = np.concatenate(cols, axis=2).astype(np.float32)
    prob = 0.55 * np.exp(-np.arange(seq_len) / 45.0) + 0.02
    y = (rs.uniform(0, 1, (batch, seq_len)) < prob).astype(np.float32)
    for i in range(batch):
        if y[i].sum() == 0:
            y[i, rs.randint(0, min(10, seq_len))] = 1.0
    if noise:
        x[:, :, 0] += (rs.standard_normal((batch, seq_len)) * noise).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(y)
