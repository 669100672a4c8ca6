"""Helpers shared by the parity tests: load a golden case, rebuild model + inputs, compare."""
import os

import numpy as np
import torch

from oracle.cases import CASE_BY_TAG, MODEL_CASES, make_criterion  # noqa: F401
from oracle.weights import fill_state_dict, synthetic_lists

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROBE = 16


def load(tag):
    return dict(np.load(os.path.join(GOLDEN, tag + ".npz"), allow_pickle=False))


def _hash_str(s):
    h = 1469598103934665603
    for ch in s.encode():
        h = ((h ^ ch) * 1099511628211) % (2 ** 63)
    return h


def probe_index(numel, key):
    rs = np.random.RandomState(abs(_hash_str(key)) % (2 ** 31))
    return rs.randint(0, numel, size=PROBE)


def build(models_mod, case, device="cpu"):
    """Instantiate case['model'] from `models_mod` (oracle.models or the HIP package's models)
    with the deterministic weights of the case; returns (model, x, y)."""
    model = getattr(models_mod, case["model"])(dropout=0.0, **case["kwargs"])
    fill_state_dict(model, case["seed"], gate_scale=case["gate_scale"], pe_scale=case.get("pe_scale"))
    x, y = synthetic_lists(case["batch"], case["seq_len"], case["n_feat"], case["seed"] + 1, noise=case.get("x_noise", 0.0))
    model = model.to(device)
    return model, x.to(device), y.to(device)


def as_list(out):
    return list(out) if isinstance(out, (list, tuple)) else [out]


def check_grads(model, gold, rtol, atol_frac):
    """Compare per-parameter gradient norm / sum / probes with the golden summary.
    Tolerance: |diff| <= rtol*|ref| + atol_frac*gnorm (probes and sums are compared relative
    to the parameter's gradient norm, which is the natural scale)."""
    worst = 0.0
    for name, prm in model.named_parameters():
        g = prm.grad
        flat = (torch.zeros_like(prm) if g is None else g).detach().reshape(-1).double().cpu()
        ref_norm = float(gold["gnorm/" + name])
        # gradients that are analytically zero (anything that only shifts all S logits of the
        # softmax head, e.g. the last LayerNorm bias) are pure rounding noise ~1e-7: floor the scale
        scale = max(ref_norm, 1e-3)
        got_norm = float(flat.norm())
        assert abs(got_norm - ref_norm) <= rtol * ref_norm + 1e-6, (name, got_norm, ref_norm)
        idx = torch.from_numpy(probe_index(flat.numel(), name))
        probes = flat[idx].numpy()
        err = np.abs(probes - gold["gprobe/" + name]).max() / scale
        assert err <= atol_frac, (name, "probe", err)
        # the sum is a cancellation-prone statistic: scale it by norm*sqrt(n)
        serr = abs(float(flat.sum()) - float(gold["gsum/" + name])) / (scale * np.sqrt(flat.numel()))
        assert serr <= atol_frac, (name, "sum", serr)
        worst = max(worst, err, serr, abs(got_norm - ref_norm) / scale)
    return worst
