"""Pin the CPU oracle to the reference: every golden vector under tests/golden/ was produced by
the reference itself (tools/make_golden.py); the oracle must reproduce it.  CPU only."""
import math

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import explicit, losses as olosses, metrics as ometrics, models as omodels
from oracle.cases import MODEL_CASES, SINGLE_CRITERIA, make_criterion

torch.set_num_threads(8)
SMALL = [c for c in MODEL_CASES if c["batch"] <= 8]


@pytest.mark.parametrize("case", MODEL_CASES, ids=lambda c: c["tag"])
def test_model_forward_loss_metrics(case):
    gold = gu.load(case["tag"])
    model, x, y = gu.build(omodels, case)
    np.testing.assert_array_equal(x.numpy(), gold["x"])
    np.testing.assert_array_equal(y.numpy(), gold["y"])
    model.train()
    outs = gu.as_list(model(x))
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().squeeze(2).numpy(), gold[f"out{i}"], rtol=0, atol=1e-6)
    k_s = ometrics.cut_positions(outs[-1].detach().numpy())
    np.testing.assert_array_equal(k_s, gold["k_s"])
    # F1: the reference divides float32 label sums; under numpy>=2 (this container) that stays
    # float32, under the numpy<1.20 it was written for it promoted to float64.  The oracle uses
    # float64, so agreement is to fp32 rounding only.
    assert abs(ometrics.Metric.f1(y.numpy(), k_s) - float(gold["f1"])) < 1e-6
    assert abs(ometrics.Metric.dcg(y.numpy(), k_s) - float(gold["dcg"])) < 1e-12
    for cname in case["criteria"]:
        loss = make_criterion(olosses, cname, case)(model(x), y)
        ref = float(gold["loss/" + cname])
        assert abs(loss.item() - ref) <= 2e-6 * max(1.0, abs(ref)), (cname, loss.item(), ref)


@pytest.mark.parametrize("case", SMALL, ids=lambda c: c["tag"])
def test_model_gradients(case):
    gold = gu.load(case["tag"])
    model, x, y = gu.build(omodels, case)
    model.train()
    loss = make_criterion(olosses, case["grad_crit"], case)(model(x), y)
    model.zero_grad()
    loss.backward()
    gu.check_grads(model, gold, rtol=1e-4, atol_frac=1e-4)


def test_losses_edge_rows_and_gradients():
    gold = gu.load("losses_edge_s300")
    y = torch.from_numpy(gold["y"])
    for cname in SINGLE_CRITERIA:
        lg = torch.from_numpy(gold["logits"]).clone().requires_grad_(True)
        p = torch.softmax(lg, dim=1).unsqueeze(2)
        p.retain_grad()
        loss = make_criterion(olosses, cname)(p, y)
        loss.backward()
        ref = float(gold["loss/" + cname])
        assert abs(loss.item() - ref) <= 2e-6 * max(1.0, abs(ref)), cname
        scale = np.abs(gold["dp/" + cname]).max()
        assert np.abs(p.grad.squeeze(2).numpy() - gold["dp/" + cname]).max() <= 2e-5 * scale, cname
        assert np.abs(lg.grad.numpy() - gold["dlogit/" + cname]).max() <= 1e-6, cname


def test_reward_matrix_closed_form_matches_loop_and_reference():
    gold = gu.load("losses_edge_s300")
    y = torch.from_numpy(gold["y"])
    for metric, tol in (("f1", 2e-7), ("dcg", 2e-5)):
        closed = olosses.reward_matrix(y, metric).numpy()
        assert np.abs(closed - gold["reward/" + metric]).max() <= tol
    # the loop-faithful builder on a few rows (it is O(S^2) python)
    rows = y[[0, 1, 2, 3, 5]][:, :120]
    for metric, tol in (("f1", 2e-7), ("dcg", 2e-5)):
        assert (olosses.reward_matrix_loop(rows, metric) - olosses.reward_matrix(rows, metric)).abs().max() <= tol


@pytest.mark.parametrize("tag,nt", [("t3", 3), ("t21", 2.1), ("t22", 2.2)])
@pytest.mark.parametrize("metric", ["f1", "dcg"])
def test_multitask_criterion(tag, nt, metric):
    gold = gu.load("losses_edge_s300")
    y = torch.from_numpy(gold["y"])
    lg = torch.from_numpy(gold["logits"]).clone().requires_grad_(True)
    cl = torch.from_numpy(gold["cls_logit"]).clone().requires_grad_(True)
    rr = torch.from_numpy(gold["rerank"]).clone().requires_grad_(True)
    p, c, r3 = torch.softmax(lg, 1).unsqueeze(2), torch.sigmoid(cl).unsqueeze(2), rr.unsqueeze(2)
    outs = [c, r3, p] if nt == 3 else ([c, p] if nt == 2.1 else [r3, p])
    loss = olosses.MtCutLoss(metric=metric, rerank_weight=0.4, classi_weight=0.6, num_tasks=nt)(outs, y)
    loss.backward()
    key = f"mtcut_{tag}_{metric}"
    assert abs(loss.item() - float(gold["loss/" + key])) <= 2e-6
    assert np.abs(lg.grad.numpy() - gold["dlogit/" + key]).max() <= 1e-6
    if nt != 2.2:
        assert np.abs(cl.grad.numpy() - gold["dcls_logit/" + key]).max() <= 1e-7
    if nt != 2.1:
        assert np.abs(rr.grad.numpy() - gold["drerank/" + key]).max() <= 1e-7     # hinge is active here


def test_rerank_edges():
    gold = gu.load("losses_edge_s300")
    # the reference raises on a batch without positives under torch 2.10 (int tensor with
    # requires_grad, utils/losses.py:138); its evident intent - a zero loss - is what we return.
    assert math.isnan(float(gold["rerank_nopos"]))
    s = torch.from_numpy(gold["rerank"][:3]).unsqueeze(2)
    assert olosses.RerankLoss()(s, torch.zeros(3, 300)).item() == 0.0
    y = torch.from_numpy(gold["y"][4:7])
    assert float(olosses.RerankLoss()((y * 5.0).unsqueeze(2), y)) == float(gold["rerank_inactive"]) == 0.0


def test_metric_known_answers():
    gold = gu.load("losses_edge_s300")
    kat_x = np.array([[1, 0, 1], [0, 0, 1], [1, 0, 0]])
    kat_k = np.array([1, 2, 1])
    # values the reference prints for its own __main__ input (utils/metrics.py:104-109)
    assert ometrics.Metric.f1(kat_x, kat_k) == 0.5555555555555555 == float(gold["kat_f1"])
    assert ometrics.Metric.dcg(kat_x, kat_k) == 0.1230234154761809 == float(gold["kat_dcg"])
    assert abs(ometrics.Metric.f1(gold["y"], gold["k_s"]) - float(gold["metric_f1"])) < 1e-6
    assert abs(ometrics.Metric.dcg(gold["y"], gold["k_s"]) - float(gold["metric_dcg"])) < 1e-12


def test_attention_runs_over_the_list_axis():
    """lists of one mini-batch are coupled through the encoder."""
    case = gu.CASE_BY_TAG["attncut_b5_s300"]
    model, x, _ = gu.build(omodels, case)
    model.eval()
    with torch.no_grad():
        base = model(x)
        x2 = x.clone()
        x2[0] += 1.0
        moved = model(x2)
    assert (base[1:] - moved[1:]).abs().max() > 1e-7


@pytest.mark.parametrize("tag", ["attncut_b5_s300", "choopy_b5_s300"])
def test_explicit_restatement_matches_stock_modules(tag):
    case = gu.CASE_BY_TAG[tag]
    gold = gu.load(tag)
    model, x, _ = gu.build(omodels, case)
    sd = model.state_dict()
    fwd = explicit.attncut_forward if case["model"] == "AttnCut" else explicit.choopy_forward
    p32 = fwd(x, sd).squeeze(2).numpy()
    p64 = fwd(x.double(), sd).squeeze(2).numpy()
    assert np.abs(p32 - gold["out0"]).max() < 1e-6
    assert np.abs(p64 - gold["out0"]).max() < 1e-6


# ---- BiCut / BiCutLoss ------------------------------------------------
from oracle.cases import BICUT_CASES  # noqa: E402
from oracle.weights import fill_state_dict, synthetic_lists  # noqa: E402


@pytest.mark.parametrize("case", BICUT_CASES, ids=lambda c: c["tag"])
def test_bicut_model_loss_and_cut_rule(case):
    gold = gu.load(case["tag"])
    model = omodels.BiCut(dropout=0.0, **case["kwargs"])
    fill_state_dict(model, case["seed"])
    x, y = synthetic_lists(case["batch"], case["seq_len"], case["n_feat"], case["seed"] + 1)
    np.testing.assert_array_equal(x.numpy(), gold["x"])
    model.train()
    out = model(x)
    np.testing.assert_allclose(out.detach().numpy(), gold["out0"], rtol=0, atol=1e-6)
    k_s = ometrics.bicut_cut_positions(out.detach().numpy())
    np.testing.assert_array_equal(k_s, gold["k_s"])
    assert abs(ometrics.Metric.f1(y.numpy(), k_s) - float(gold["f1"])) < 1e-6
    for metric in case["criteria"]:
        o = model(x)
        o.retain_grad()
        loss = olosses.BiCutLoss(metric=metric)(o, y)
        model.zero_grad()
        loss.backward()
        ref = float(gold["loss/" + metric])
        assert abs(loss.item() - ref) <= 2e-6 * max(1.0, abs(ref)), (metric, loss.item(), ref)
        np.testing.assert_allclose(o.grad.numpy(), gold["dout/" + metric], rtol=1e-6, atol=1e-7)
        if metric == case["grad_crit"]:
            gu.check_grads(model, gold, rtol=1e-4, atol_frac=1e-4)


def test_bicutloss_edge_rows():
    gold = gu.load("bicutloss_edge_s50")
    y = torch.from_numpy(gold["y"])
    for metric in ("nci", "f1"):
        o = torch.softmax(torch.from_numpy(gold["logits"]), dim=2).requires_grad_(True)
        loss = olosses.BiCutLoss(metric=metric)(o, y)
        loss.backward()
        ref = float(gold["loss/" + metric])
        assert abs(loss.item() - ref) <= 2e-6 * max(1.0, abs(ref))
        np.testing.assert_allclose(o.grad.numpy(), gold["dout/" + metric], rtol=1e-6, atol=1e-7)
    k_s = ometrics.bicut_cut_positions(torch.softmax(torch.from_numpy(gold["logits"]), dim=2).numpy())
    np.testing.assert_array_equal(k_s, gold["k_s"])


def test_task_metrics():
    gold = gu.load("task_metrics_s300")
    assert abs(ometrics.taskr_metric(gold["y"], gold["pred"]) - float(gold["taskr"])) < 1e-9
    assert abs(ometrics.taskc_metric(gold["y"], gold["pred"]) - float(gold["taskc"])) < 1e-12
    assert abs(ometrics.taskc_metric(gold["y"], gold["pred_ties"]) - float(gold["taskc_ties"])) < 1e-12


def test_wassdist_loss_and_gradient():
    gold = gu.load("wassdist")
    for tag in sorted({k.split("/")[0] for k in gold}):
        p = torch.softmax(torch.from_numpy(gold[f"{tag}/logits"]), dim=1).unsqueeze(2).requires_grad_(True)
        loss = olosses.WassDistLoss(eps=float(gold[f"{tag}/eps"]), max_iter=100)(p, torch.from_numpy(gold[f"{tag}/y"]))
        loss.backward()
        ref = float(gold[f"{tag}/loss"])
        assert abs(loss.item() - ref) <= 1e-5 * max(1.0, abs(ref)), (tag, loss.item(), ref)
        dp = gold[f"{tag}/dp"]
        np.testing.assert_allclose(p.grad.squeeze(2).numpy(), dp, rtol=0, atol=1e-4 * np.abs(dp).max())


def test_dcg_penalty_argument_against_reference():
    """Metric.dcg(..., penalty) / Metric_for_Loss.dcg(..., penalty) (utils/metrics.py:27,94) at non-default penalties,
    values produced by the reference (tools/make_golden.py losses)."""
    import numpy as np
    import torch
    from oracle import losses as ol, metrics as om
    gold = gu.load("losses_edge_s300")
    y = torch.from_numpy(gold["y"])
    for pen in (-0.5, -2.0, 0.25):
        assert abs(om.Metric.dcg(gold["y"], gold["k_s"], pen) - float(gold[f"metric_dcg_pen/{pen:g}"])) < 1e-9
        want = torch.from_numpy(gold[f"reward_dcg_pen/{pen:g}"])
        rows = y[[0, 1, 4, 7]]
        assert float((ol.reward_matrix(rows, "dcg", pen) - want).abs().max()) < 4e-5
        loop = torch.stack([torch.stack([ol.reward_dcg(r, j + 1, pen) for j in range(0, 300, 37)]) for r in rows])
        assert float((loop - want[:, 0:300:37]).abs().max()) < 1e-6


def test_explicit_encoder_layer_dropout_masks_match_the_stock_module(monkeypatch):
    """oracle/explicit.py's encoder layer with the four train-mode dropouts given as keep-masks (the reference for the
    GPU's dropout tests, where the device's own masks are exported as data) against the STOCK nn.TransformerEncoderLayer
    in train() with torch's dropouts replaced by those same masks: dropout1 / dropout / dropout2 are the module's own
    nn.Dropout children (swapped for mask multipliers), the attention-probability dropout is F.dropout inside
    F.multi_head_attention_forward's need_weights path (patched to multiply by the mask)."""
    import torch.nn.functional as F
    torch.manual_seed(7)
    B, S, E, H, FF, p = 9, 4, 32, 4, 64, 0.3
    layer = torch.nn.TransformerEncoderLayer(d_model=E, nhead=H, dim_feedforward=FF, dropout=p).double()
    layer.train()
    keep = lambda *shape: (torch.rand(*shape) >= p).double() / (1 - p)
    masks = {"attn": keep(S, H, B, B), "res1": keep(B, S, E), "ffn": keep(B, S, FF), "res2": keep(B, S, E)}

    class Mask(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x):
            return x * self.m

    layer.dropout1, layer.dropout, layer.dropout2 = Mask(masks["res1"]), Mask(masks["ffn"]), Mask(masks["res2"])
    # the attention weights F.dropout sees are (S*H, B, B) in (position, head) order
    monkeypatch.setattr(F, "dropout", lambda w, p=0.5, training=True, inplace=False: w * masks["attn"].reshape(w.shape))
    x = torch.randn(B, S, E, dtype=torch.float64, requires_grad=True)
    # need_weights=True takes the explicit softmax -> dropout -> bmm path (the SDPA path draws its own mask internally)
    att = layer.self_attn(x, x, x, need_weights=True)[0]
    want = layer.norm1(x + layer.dropout1(att))
    want = layer.norm2(want + layer.dropout2(layer.linear2(layer.dropout(torch.relu(layer.linear1(want))))))
    monkeypatch.undo()
    x2 = x.detach().clone().requires_grad_(True)
    got = explicit.encoder_layer(x2, layer.state_dict(), "", H, masks=masks)
    assert float((got - want).abs().max()) < 1e-12
    g = torch.randn(B, S, E, dtype=torch.float64)
    want.backward(g)
    got.backward(g)
    assert float((x2.grad - x.grad).abs().max()) < 1e-11
    # and without masks it is the eval() / dropout-0 layer
    layer2 = torch.nn.TransformerEncoderLayer(d_model=E, nhead=H, dim_feedforward=FF, dropout=0.0).double()
    layer2.load_state_dict({k: v for k, v in layer.state_dict().items()})
    assert float((explicit.encoder_layer(x, layer.state_dict(), "", H) - layer2(x)).abs().max()) < 1e-12
