#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in this container.

Only ever run in the build container (where /root/reference is mounted read-only).  The
reference's classes are imported unmodified; the one accommodation is a stub module for
`numpy.lib.financial` (an unused import at utils/metrics.py:3 that modern numpy dropped).
Nothing from the reference is written into this repository: the fixtures hold inputs and
the outputs the reference produced for them (data), weights are regenerated on every side
from `oracle/weights.py`.

    python tools/make_golden.py            # regenerates every tests/golden/*.npz
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("RLT_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, REPO)
from oracle.weights import fill_state_dict, synthetic_lists  # noqa: E402
from oracle.cases import BICUT_CASES, MODEL_CASES, SINGLE_CRITERIA, make_criterion  # noqa: E402

# ---- import the reference (models/, utils/) -------------------------------------------------
_stub = types.ModuleType("numpy.lib.financial")
_stub.irr = None
sys.modules["numpy.lib.financial"] = _stub
sys.path.insert(0, REF)
import models as ref_models  # noqa: E402
from utils import losses as ref_losses  # noqa: E402
from utils.metrics import Metric as RefMetric, Metric_for_Loss as RefMetricForLoss  # noqa: E402
sys.path.remove(REF)

torch.set_num_threads(8)
PROBE = 16


def probe_index(numel, key):
    rs = np.random.RandomState(abs(hash_str(key)) % (2 ** 31))
    return rs.randint(0, numel, size=PROBE)


def hash_str(s):
    h = 1469598103934665603
    for ch in s.encode():
        h = ((h ^ ch) * 1099511628211) % (2 ** 63)
    return h


def grad_summary(model):
    out = {}
    for name, prm in model.named_parameters():
        g = prm.grad
        if g is None:
            g = torch.zeros_like(prm)
        flat = g.detach().reshape(-1).double()
        out["gnorm/" + name] = np.float64(flat.norm().item())
        out["gsum/" + name] = np.float64(flat.sum().item())
        out["gprobe/" + name] = flat[torch.from_numpy(probe_index(flat.numel(), name))].numpy()
    return out


def as_list(out):
    return list(out) if isinstance(out, (list, tuple)) else [out]


def run_model_case(case):
    tag, batch, seq_len = case["tag"], case["batch"], case["seq_len"]
    cls = getattr(ref_models, case["model"])
    model = cls(dropout=0.0, **case["kwargs"])
    fill_state_dict(model, case["seed"], gate_scale=case["gate_scale"], pe_scale=case.get("pe_scale"))
    x, y = synthetic_lists(batch, seq_len, case["n_feat"], case["seed"] + 1, noise=case.get("x_noise", 0.0))
    gs = case["gate_scale"]
    rec = {"x": x.numpy(), "y": y.numpy(), "seed": np.int64(case["seed"]),
           "gate_scale": np.float64(-1.0 if gs is None else gs)}

    model.train()
    outs = as_list(model(x))
    for i, o in enumerate(outs):
        rec[f"out{i}"] = o.detach().squeeze(2).numpy()
    model.eval()
    with torch.no_grad():
        outs_eval = as_list(model(x))
    rec["eval_max_abs_diff"] = np.float64(max((a.detach() - b).abs().max().item() for a, b in zip(outs, outs_eval)))

    cut = outs[-1].detach().squeeze(2).numpy()
    k_s = np.argmax(cut, axis=1) + 1
    rec["k_s"] = k_s.astype(np.int64)
    rec["f1"] = np.float64(RefMetric.f1(y.numpy(), k_s))
    rec["dcg"] = np.float64(RefMetric.dcg(y.numpy(), k_s))

    model.train()
    for cname in case["criteria"]:
        crit = make_criterion(ref_losses, cname, case)
        out = model(x)
        loss = crit(out, y)
        rec["loss/" + cname] = np.float64(loss.item())
        if cname == case["grad_crit"]:
            model.zero_grad()
            loss.backward()
            rec.update(grad_summary(model))
            rec["grad_crit"] = np.array(cname)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **rec)
    print(f"{tag}: f1={rec['f1']:.6f} dcg={rec['dcg']:.6f} "
          + " ".join(f"{k[5:]}={float(v):.6f}" for k, v in rec.items() if k.startswith("loss/")), flush=True)


def single_criteria():
    return {name: (lambda name=name: make_criterion(ref_losses, name)) for name in SINGLE_CRITERIA}


def model_cases(only=None):
    for case in MODEL_CASES:
        if only is None or case["tag"] in only:
            run_model_case(case)


def loss_cases():
    """Loss-only vectors incl. the edge rows; stores dL/dp in full."""
    rs = np.random.RandomState(7)
    n_pos = 300
    y = (rs.uniform(0, 1, (10, n_pos)) < (0.5 * np.exp(-np.arange(n_pos) / 40.0) + 0.02)).astype(np.float32)
    y[0] = 0.0                       # no relevant document
    y[1] = 1.0                       # all relevant
    y[2] = 0.0; y[2, 0] = 1.0        # single positive at the top
    y[3] = 0.0; y[3, n_pos - 1] = 1.0  # single positive at the bottom
    logits = rs.standard_normal((10, n_pos)).astype(np.float32) * 2.0
    rec = {"y": y, "logits": logits}
    y_t = torch.from_numpy(y)
    for cname, make in single_criteria().items():
        lg = torch.from_numpy(logits).clone().requires_grad_(True)
        p = torch.softmax(lg, dim=1).unsqueeze(2)
        p.retain_grad()
        loss = make()(p, y_t)
        loss.backward()
        rec["loss/" + cname] = np.float64(loss.item())
        rec["dp/" + cname] = p.grad.squeeze(2).numpy()
        rec["dlogit/" + cname] = lg.grad.numpy()
    # multi-task criterion on synthetic heads
    cls_logit = rs.standard_normal((10, n_pos)).astype(np.float32)
    rr = (rs.standard_normal((10, n_pos)) - 0.3 * y).astype(np.float32)   # relevant docs score lower: hinge active
    rec["cls_logit"], rec["rerank"] = cls_logit, rr
    for nt, tag in ((3, "t3"), (2.1, "t21"), (2.2, "t22")):
        for metric in ("f1", "dcg"):
            lg = torch.from_numpy(logits).clone().requires_grad_(True)
            cl = torch.from_numpy(cls_logit).clone().requires_grad_(True)
            r_ = torch.from_numpy(rr).clone().requires_grad_(True)
            p = torch.softmax(lg, dim=1).unsqueeze(2)
            c = torch.sigmoid(cl).unsqueeze(2)
            r3 = r_.unsqueeze(2)
            outs = [c, r3, p] if nt == 3 else ([c, p] if nt == 2.1 else [r3, p])
            loss = ref_losses.MtCutLoss(metric=metric, rerank_weight=0.4, classi_weight=0.6, num_tasks=nt)(outs, y_t)
            loss.backward()
            key = f"mtcut_{tag}_{metric}"
            rec["loss/" + key] = np.float64(loss.item())
            rec["dlogit/" + key] = lg.grad.numpy()
            if cl.grad is not None:
                rec["dcls_logit/" + key] = cl.grad.numpy()
            if r_.grad is not None:
                rec["drerank/" + key] = r_.grad.numpy()
    # rerank hinge edge: a batch without positives -> exactly 0
    z = torch.zeros(3, n_pos)
    s = torch.from_numpy(rr[:3]).clone().unsqueeze(2).requires_grad_(True)
    # (with torch 2.10 the reference raises here: `t.tensor(0, requires_grad=True)` is an int
    #  tensor, utils/losses.py:138; recorded as NaN = "reference raised")
    try:
        rec["rerank_nopos"] = np.float64(ref_losses.RerankLoss()(s, z).item())
    except RuntimeError as exc:
        print("reference RerankLoss raised on a batch without positives:", exc)
        rec["rerank_nopos"] = np.float64("nan")
    # rerank hinge inactive: positives already score much higher
    s2 = (torch.from_numpy(y[4:7]) * 5.0).unsqueeze(2)
    rec["rerank_inactive"] = np.float64(float(ref_losses.RerankLoss()(s2, torch.from_numpy(y[4:7]))))
    # reward matrices themselves (the B*S python loop of the reference), via ChoopyLoss linearity:
    # r[i][j] = -B * dLoss/dp[i][j]
    for metric in ("f1", "dcg"):
        rec["reward/" + metric] = -10.0 * rec["dp/choopy_" + metric]
    # evaluation metrics at a spread of cut positions + the reference's own known-answer input
    k_s = rs.randint(1, n_pos + 1, size=10)
    k_s[0], k_s[1] = 1, n_pos
    rec["k_s"] = k_s.astype(np.int64)
    rec["metric_f1"] = np.float64(RefMetric.f1(y, k_s))
    rec["metric_dcg"] = np.float64(RefMetric.dcg(y, k_s))
    kat_x = np.array([[1, 0, 1], [0, 0, 1], [1, 0, 0]])
    kat_k = np.array([1, 2, 1])
    rec["kat_f1"] = np.float64(RefMetric.f1(kat_x, kat_k))
    rec["kat_dcg"] = np.float64(RefMetric.dcg(kat_x, kat_k))
    # the `penalty` argument of Metric.dcg (utils/metrics.py:27) and Metric_for_Loss.dcg (:94), non-default values
    for pen in (-0.5, -2.0, 0.25):
        rec[f"metric_dcg_pen/{pen:g}"] = np.float64(RefMetric.dcg(y, k_s, penalty=pen))
        r = np.zeros((4, n_pos), dtype=np.float32)
        for i, row in enumerate((0, 1, 4, 7)):                      # no relevant doc, all relevant, two random rows
            for j in range(n_pos):
                r[i, j] = float(RefMetricForLoss.dcg(y_t[row], j + 1, penalty=pen))
        rec[f"reward_dcg_pen/{pen:g}"] = r
    np.savez_compressed(os.path.join(OUT, "losses_edge_s300.npz"), **rec)
    print("losses_edge_s300: kat_f1=%.16f kat_dcg=%.16f" % (rec["kat_f1"], rec["kat_dcg"]), flush=True)



def bicut_cases():
    """BiCut + BiCutLoss (models/Bicut.py, utils/losses.py:11-45) and the cut rule of run.py:131-136, all run by the
    reference; outputs are (B,S,2)."""
    for case in BICUT_CASES:
        model = ref_models.BiCut(dropout=0.0, **case["kwargs"])
        fill_state_dict(model, case["seed"])
        x, y = synthetic_lists(case["batch"], case["seq_len"], case["n_feat"], case["seed"] + 1)
        rec = {"x": x.numpy(), "y": y.numpy(), "seed": np.int64(case["seed"])}
        model.train()
        out = model(x)
        rec["out0"] = out.detach().numpy()
        pred = np.argmax(out.detach().numpy(), axis=2)                      # run.py:131-136
        S = case["seq_len"]
        k_s = np.array([S if r.sum() == S else np.argmin(r) + 1 for r in pred], dtype=np.int64)
        rec["k_s"] = k_s
        rec["f1"] = np.float64(RefMetric.f1(y.numpy(), k_s))
        rec["dcg"] = np.float64(RefMetric.dcg(y.numpy(), k_s))
        for metric in case["criteria"]:
            crit = ref_losses.BiCutLoss(metric=metric)
            o = model(x)
            o.retain_grad()
            loss = crit(o, y)
            rec["loss/" + metric] = np.float64(loss.item())
            model.zero_grad()
            loss.backward()
            rec["dout/" + metric] = o.grad.numpy()
            if metric == case["grad_crit"]:
                rec.update(grad_summary(model))
        np.savez_compressed(os.path.join(OUT, case["tag"] + ".npz"), **rec)
        print(f"{case['tag']}: k_s={k_s.tolist()} f1={rec['f1']:.6f} "
              + " ".join(f"{m}={float(rec['loss/' + m]):.6f}" for m in case["criteria"]), flush=True)
    # loss-only edge rows: every position continue / every position truncate / exact ties / random
    rs = np.random.RandomState(173)
    B, S = 7, 50
    lg = rs.standard_normal((B, S, 2)).astype(np.float32)
    lg[0, :, 1] = lg[0, :, 0] + 1.0          # all continue -> nothing masked
    lg[1, :, 0] = lg[1, :, 1] + 1.0          # all truncate
    lg[2, :, 1] = lg[2, :, 0]                # ties -> argmax picks class 0
    lg[3, :-1, 1] = lg[3, :-1, 0] + 1.0      # only the last position truncates
    lg[3, -1, 0] = lg[3, -1, 1] + 1.0
    lg[4, 1:, 1] = lg[4, 1:, 0] + 1.0        # only the first position truncates
    lg[4, 0, 0] = lg[4, 0, 1] + 1.0
    y = (rs.uniform(0, 1, (B, S)) < 0.2).astype(np.float32)
    y[5] = 0.0
    y[6] = 1.0
    rec = {"logits": lg, "y": y}
    for metric in ("nci", "f1"):
        o = torch.softmax(torch.from_numpy(lg), dim=2).requires_grad_(True)
        loss = ref_losses.BiCutLoss(metric=metric)(o, torch.from_numpy(y))
        loss.backward()
        rec["loss/" + metric] = np.float64(loss.item())
        rec["dout/" + metric] = o.grad.numpy()
    pred = np.argmax(torch.softmax(torch.from_numpy(lg), dim=2).numpy(), axis=2)
    rec["k_s"] = np.array([S if r.sum() == S else np.argmin(r) + 1 for r in pred], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "bicutloss_edge_s50.npz"), **rec)
    print("bicutloss_edge_s50:", {m: float(rec["loss/" + m]) for m in ("nci", "f1")}, rec["k_s"].tolist(), flush=True)



def task_metric_case():
    """Metric.taskr_metric / Metric.taskc_metric (utils/metrics.py:40-76) run by the reference (sklearn's AUC)."""
    rs = np.random.RandomState(181)
    B, S = 9, 300
    pred = rs.standard_normal((B, S)).astype(np.float32)
    y = (rs.uniform(0, 1, (B, S)) < 0.15).astype(np.float32)
    y[2] = 0.0                                   # single-class lists: skipped by taskc_metric
    y[5] = 1.0
    pred_ties = pred.copy()
    pred_ties[7] = np.round(pred_ties[7], 1)     # ties inside a list: AUC counts them 1/2 (the DCG of tied documents
    rec = {"pred": pred, "pred_ties": pred_ties, "y": y,        # depends on numpy's unstable argsort: not pinned)
           "taskr": np.float64(RefMetric.taskr_metric(y, pred)),
           "taskc": np.float64(RefMetric.taskc_metric(y, pred)),
           "taskc_ties": np.float64(RefMetric.taskc_metric(y, pred_ties))}
    np.savez_compressed(os.path.join(OUT, "task_metrics_s300.npz"), **rec)
    print("task_metrics_s300:", float(rec["taskr"]), float(rec["taskc"]), flush=True)



def wass_cases():
    """WassDistLoss (utils/losses.py:236-311) run by the reference: loss and d(loss)/d(output) for a few (B, S, eps)."""
    rec = {}
    for tag, B, S, eps, seed in [("b6_s40_e3", 6, 40, 1e-3, 191), ("b6_s40_e2", 6, 40, 1e-2, 192), ("b17_s300_e2", 17, 300, 1e-2, 193),
                                 ("b33_s100_e3", 33, 100, 1e-3, 194)]:
        rs = np.random.RandomState(seed)
        lg = rs.standard_normal((B, S)).astype(np.float32)
        y = (rs.uniform(0, 1, (B, S)) < 0.2).astype(np.float32)
        p = torch.softmax(torch.from_numpy(lg), dim=1).unsqueeze(2).requires_grad_(True)
        loss = ref_losses.WassDistLoss(eps=eps, max_iter=100)(p, torch.from_numpy(y))
        loss.backward()
        rec[f"{tag}/logits"], rec[f"{tag}/y"] = lg, y
        rec[f"{tag}/eps"] = np.float64(eps)
        rec[f"{tag}/loss"] = np.float64(loss.item())
        rec[f"{tag}/dp"] = p.grad.squeeze(2).numpy()
        print("wass", tag, float(loss.item()), float(np.abs(rec[f"{tag}/dp"]).max()), flush=True)
    np.savez_compressed(os.path.join(OUT, "wassdist.npz"), **rec)


def data_case():
    """Run the reference's OWN loaders (dataloader/attncut_dataloader.py, choopy_dataloader.py) on a small
    synthetic robust04-format pickle set written by our generator; store the tensors they produce."""
    import tempfile
    sys.path.insert(0, os.path.join(REPO, "ranked-list-truncation_amd"))
    from dataloader.synth import write_synthetic_robust04
    sys.path.remove(os.path.join(REPO, "ranked-list-truncation_amd"))
    for mod in [m for m in sys.modules if m == "dataloader" or m.startswith("dataloader.")]:
        del sys.modules[mod]
    sys.path.insert(0, REF)
    import dataloader.attncut_dataloader as ref_at
    import dataloader.choopy_dataloader as ref_cp
    sys.path.remove(REF)
    with tempfile.TemporaryDirectory() as tmp:
        write_synthetic_robust04(tmp, "robust04", "drmm_tks", n_train=7, n_test=3, seq_len=300, seed=77)
        ref_at.DATASET_BASE = tmp
        ref_cp.DATASET_BASE = tmp
        a = ref_at.Rank_Dataset("robust04", "drmm_tks")
        c = ref_cp.Rank_Dataset("robust04", "drmm_tks")
        rec = {"at_X_train": a.getX_train().numpy(), "at_X_test": a.getX_test().numpy(),
               "at_y_train": a.gety_train().numpy(), "at_y_test": a.gety_test().numpy(),
               "cp_X_train": c.getX_train().numpy(), "cp_X_test": c.getX_test().numpy(),
               "cp_y_train": c.gety_train().numpy(), "cp_y_test": c.gety_test().numpy()}
    np.savez_compressed(os.path.join(OUT, "dataloader_synth77.npz"), **rec)
    print("dataloader_synth77:", {k: v.shape for k, v in rec.items()}, flush=True)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1] if len(sys.argv) > 1 else "all"
    if only in ("all", "losses"):
        loss_cases()
    if only in ("all", "data"):
        data_case()
    if only in ("all", "models"):
        model_cases(set(sys.argv[2:]) or None)
    if only in ("all", "bicut"):
        bicut_cases()
    if only in ("all", "taskmetrics"):
        task_metric_case()
    if only in ("all", "wass"):
        wass_cases()
