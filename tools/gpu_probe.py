#!/usr/bin/env python3
"""Developer probe (GPU box): run every HIP op against a torch-CPU / oracle reference and print the
errors WITHOUT stopping at the first failure.  `python tools/gpu_probe.py [section ...]`.
Not part of the test suite (tests/ holds the asserted versions)."""
import math
import os
import sys
import time
import traceback

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "ranked-list-truncation_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np
import torch
import torch.nn.functional as F

from rlt_hip import native as N
from rlt_hip import ops

dev = torch.device("cuda")
torch.manual_seed(0)
RESULTS = []


def mfma_tol(tol):
    """Op-level tolerance of an MFMA contraction: as written for the exact-fp32 mode; the split-bf16 mode carries
    ~2^-16 relative error per product, so its op-level bound is 6x looser (model-level bounds are not scaled)."""
    return tol * (6.0 if N.get_precision() == "bf16x3" else 1.0)


def report(name, err, tol):
    ok = err <= tol and not math.isnan(err)
    RESULTS.append((name, err, tol, ok))
    print(f"{'OK  ' if ok else 'FAIL'} {name:58s} err={err:.3e} tol={tol:.1e}", flush=True)


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def section(fn):
    fn._is_section = True
    return fn


# ------------------------------------------------------------------------------------------------
@section
def gemm():
    for (ta, tb, M, Nn, K) in [(0, 1, 300, 768, 256), (0, 1, 1500, 1024, 3), (0, 0, 1500, 256, 2048), (1, 0, 2048, 256, 1500),
                               (0, 1, 129, 130, 17), (1, 1, 70, 33, 45), (1, 0, 512, 128, 40000), (0, 0, 257, 3, 512),
                               (1, 0, 512, 3, 9000)]:
        A = torch.randn((K, M) if ta else (M, K))
        B = torch.randn((Nn, K) if tb else (K, Nn))
        bias = torch.randn(Nn)
        ref = (A.t() if ta else A).double() @ (B.t() if tb else B).double() + bias.double()
        C = torch.empty(M, Nn, device=dev)
        Ad, Bd, bd = A.to(dev), B.to(dev), bias.to(dev)
        ops.gemm(ta, tb, M, Nn, K, Ad, A.shape[1], Bd, B.shape[1], C, Nn, bias=bd)
        report(f"gemm ta={ta} tb={tb} {M}x{Nn}x{K}", rel(C, ref), mfma_tol(2e-6 * math.sqrt(K) + 1e-6))
    # 256-aligned shapes take the 256x256-tile kernel in split-bf16 mode (all four layouts, split-K, fused side sums)
    for (ta, tb, M, Nn, K) in [(0, 1, 512, 256, 64), (0, 0, 256, 512, 96), (1, 0, 512, 256, 16384), (1, 1, 256, 256, 8192),
                               (0, 1, 1024, 768, 256)]:
        A = torch.randn((K, M) if ta else (M, K))
        B = torch.randn((Nn, K) if tb else (K, Nn))
        bias = torch.randn(Nn)
        ref = (A.t() if ta else A).double() @ (B.t() if tb else B).double() + bias.double()
        C = torch.empty(M, Nn, device=dev)
        cs = torch.empty(M, device=dev) if ta else None
        ops.gemm(ta, tb, M, Nn, K, A.to(dev), A.shape[1], B.to(dev), B.shape[1], C, Nn, bias=bias.to(dev), colsum_a=cs)
        report(f"gemm256 ta={ta} tb={tb} {M}x{Nn}x{K}", rel(C, ref), mfma_tol(2e-6 * math.sqrt(K) + 1e-6))
        if ta:
            report(f"gemm256 colsum ta={ta} tb={tb} {M}x{Nn}x{K}", rel(cs, A.double().sum(0)), 1e-5)
    A, B, C0 = torch.randn(512, 64), torch.randn(64, 256), torch.randn(512, 256)
    C = C0.clone().to(dev)
    ops.gemm(0, 0, 512, 256, 64, A.to(dev), 64, B.to(dev), 256, C, 256, flags=N.GEMM_ACCUMULATE)
    report("gemm256 accumulate", rel(C, A.double() @ B.double() + C0.double()), mfma_tol(1e-5))
    # relu + accumulate
    A, B = torch.randn(200, 64), torch.randn(96, 64)
    C0 = torch.randn(200, 96)
    C = C0.clone().to(dev)
    ops.gemm(0, 1, 200, 96, 64, A.to(dev), 64, B.to(dev), 64, C, 96, flags=N.GEMM_RELU | N.GEMM_ACCUMULATE)
    report("gemm relu+accumulate", rel(C, torch.relu(A @ B.t() + C0)), mfma_tol(1e-5))
    # fused side products: ReLU mask on the output, column sums of A^T (bias gradient) with and without split-K
    for (M, Nn, K) in [(300, 200, 77), (512, 3, 9000), (768, 256, 50000)]:
        A, B = torch.randn(K, M), torch.randn(K, Nn)
        mask = torch.randn(M, Nn)
        C = torch.empty(M, Nn, device=dev)
        cs = torch.empty(M, device=dev)
        ops.gemm(1, 0, M, Nn, K, A.to(dev), M, B.to(dev), Nn, C, Nn, relu_mask=mask.to(dev), ldmask=Nn, colsum_a=cs)
        ref = (A.t().double() @ B.double()) * (mask > 0)
        report(f"gemm_ex mask {M}x{Nn}x{K}", rel(C, ref), mfma_tol(2e-6 * math.sqrt(K) + 1e-6))
        report(f"gemm_ex colsum {M}x{Nn}x{K}", rel(cs, A.double().sum(0)), 1e-5)
    # K = 256 with >= 8192 rows: the weights-stationary streaming kernel of the bf16x6 mode (csrc/gemm6s.hip) - ragged last block, rows
    # behind the end untouched, one / three / four panels, both layouts of the weight, two biases, strided A and C
    for (M, Nn, K) in [(8209, 256, 256), (8192 + 31, 768, 256), (12000, 1024, 256), (8192 + 7, 512, 128), (20000, 2048, 128), (8192 + 19, 384, 128), (9000, 128, 128)]:
        A, W, W2, b1, b2 = torch.randn(M, K + 4), torch.randn(Nn, K) / 16, torch.randn(K, Nn) / 16, torch.randn(Nn), torch.randn(Nn)
        for tb, Bm in ((1, W), (0, W2)):
            C = torch.full((M + 3, Nn + 4), 7.0, device=dev)
            ops.gemm(0, tb, M, Nn, K, A.to(dev), K + 4, Bm.to(dev), Bm.shape[1], C, Nn + 4, bias=b1.to(dev), bias2=b2.to(dev))
            ref = A[:, :K].double() @ (Bm.t() if tb else Bm).double() + b1.double() + b2.double()
            report(f"gemm K={K} stream tb={tb} {M}x{Nn}", rel(C[:M, :Nn], ref), mfma_tol(2e-6 * 16 + 1e-6))
            report(f"gemm K={K} stream tb={tb} {M}x{Nn}: rows / columns outside C untouched",
                   float((C[M:] != 7.0).sum() + (C[:, Nn:] != 7.0).sum()), 0)
    # 1-bit ReLU mask pair (rlt_gemm_bits): interior and edge tiles; the last two shapes take the streaming kernel in bf16x6 mode
    for (M, Nn, K) in [(256, 128, 64), (300, 96, 40), (1000, 2048, 256), (512, 512, 128), (8192 + 45, 512, 256), (16384, 256, 256), (8192 + 45, 512, 128)]:
        A, W, bias = torch.randn(M, K), torch.randn(Nn, K), torch.randn(Nn)
        H = torch.empty(M, Nn, device=dev)
        bits = ops.alloc_relu_bits(M, Nn, dev)
        ops.gemm_bits(0, 1, M, Nn, K, A.to(dev), K, W.to(dev), K, H, Nn, bias=bias.to(dev), flags=N.GEMM_RELU, bits_out=bits)
        ref = torch.relu(A.double() @ W.double().t() + bias.double())
        report(f"gemm_bits relu fwd {M}x{Nn}x{K}", rel(H, ref), mfma_tol(1e-5))
        unpacked = ops.unpack_relu_bits(bits, M).cpu()
        report(f"gemm_bits mask bits {M}x{Nn}x{K}", float((unpacked != (H.cpu() > 0)).sum()), 0)
        dY, W2 = torch.randn(M, K), torch.randn(K, Nn)
        dH = torch.empty(M, Nn, device=dev)
        ops.gemm_bits(0, 0, M, Nn, K, dY.to(dev), K, W2.to(dev), Nn, dH, Nn, bits_in=bits, mask_scale=1.25)
        refd = (dY.double() @ W2.double()) * (H.cpu() > 0) * 1.25
        report(f"gemm_bits masked bwd {M}x{Nn}x{K}", rel(dH, refd), mfma_tol(2e-6 * math.sqrt(K) + 1e-6))
        # same product with the train-mode dropout of the FFN hidden fused in: output = relu * keep-mask of
        # rlt_dropout_mask (the un-hoisted form of the same row x column hash), bits = what survived both
        pd, sd = 0.4, 991 + M
        mk = torch.empty(M, Nn, device=dev)
        N.call("rlt_dropout_mask", sd, M, Nn, pd, N.ptr(mk), N.stream())
        Hd = torch.empty(M, Nn, device=dev)
        bits.zero_()
        ops.gemm_bits(0, 1, M, Nn, K, A.to(dev), K, W.to(dev), K, Hd, Nn, bias=bias.to(dev), flags=N.GEMM_RELU, bits_out=bits,
                      drop_p=pd, seed=sd)
        # (K = 256 with >= 8192 rows: in bf16x6 mode the dropout-free product ran on the streaming kernel, this one on the tiled kernel:
        #  two roundings of the same sum)
        report(f"gemm_bits relu+dropout fwd {M}x{Nn}x{K}", rel(Hd, H.double() * mk.double()), 1e-6 if M < 8192 else 5e-6)       # (K = 128 alike)
        unpacked = ops.unpack_relu_bits(bits, M).cpu()
        report(f"gemm_bits relu+dropout bits {M}x{Nn}x{K}", float((unpacked != (Hd.cpu() > 0)).sum()), 0)
        report(f"gemm_bits dropout keep fraction {M}x{Nn}x{K}", abs(float((mk > 0).float().mean()) - (1 - pd)), 8e-3)
    X = torch.randn(5000, 300)
    out = torch.empty(300, device=dev)
    ops.colsum(X.to(dev), 300, 5000, 300, out)
    report("colsum 5000x300", rel(out, X.double().sum(0)), 1e-5)


@section
def losses():
    import golden_util as gu
    from oracle import losses as ol
    from oracle.cases import SINGLE_CRITERIA, make_criterion
    from utils import losses as hl
    gold = gu.load("losses_edge_s300")
    y = torch.from_numpy(gold["y"])
    for cname in SINGLE_CRITERIA:
        lg = torch.from_numpy(gold["logits"]).clone()
        p = torch.softmax(lg, 1).unsqueeze(2).to(dev).requires_grad_(True)
        loss = make_criterion(hl, cname)(p, y.to(dev))
        loss.backward()
        ref = float(gold["loss/" + cname])
        report(f"loss {cname}", abs(loss.item() - ref) / max(1, abs(ref)), 1e-5)
        scale = np.abs(gold["dp/" + cname]).max()
        report(f"dloss/dp {cname}", float(np.abs(p.grad.squeeze(2).cpu().numpy() - gold["dp/" + cname]).max() / scale), 1e-4)
    wg = gu.load("wassdist")
    for tag in sorted({k.split("/")[0] for k in wg}):
        p = torch.softmax(torch.from_numpy(wg[f"{tag}/logits"]), dim=1).unsqueeze(2).to(dev).requires_grad_(True)
        loss = hl.WassDistLoss(eps=float(wg[f"{tag}/eps"]), max_iter=100)(p, torch.from_numpy(wg[f"{tag}/y"]).to(dev))
        loss.backward()
        ref = float(wg[f"{tag}/loss"])
        report(f"wassdist loss {tag}", abs(loss.item() - ref) / max(1.0, abs(ref)), 1e-4)
        dpr = wg[f"{tag}/dp"]
        # (-C + u + v) / eps has magnitude ~1e4 at eps = 1e-3: one fp32 ulp there is ~1e-3 in the exponent, so two fp32
        # evaluation orders of the same Sinkhorn iterations (torch's and ours) differ by ~1e-3 in the plan and its gradient
        report(f"wassdist dp {tag}", float(np.abs(p.grad.squeeze(2).cpu().numpy() - dpr).max() / np.abs(dpr).max()), 5e-3)
    for metric in ("f1", "dcg"):
        r = ops.reward_matrix(y.to(dev), N.METRIC_F1 if metric == "f1" else N.METRIC_DCG)
        report(f"reward matrix {metric}", float(np.abs(r.cpu().numpy() - gold["reward/" + metric]).max()), 2e-5)
    # the `penalty` argument of Metric_for_Loss.dcg / Metric.dcg (utils/metrics.py:94,27), reference values
    from utils.metrics import Metric, Metric_for_Loss
    for pen in (-0.5, -2.0, 0.25):
        r = ops.reward_matrix(y[[0, 1, 4, 7]].to(dev), N.METRIC_DCG, penalty=pen)
        report(f"reward matrix dcg penalty {pen:g}", float(np.abs(r.cpu().numpy() - gold[f"reward_dcg_pen/{pen:g}"]).max()), 4e-5)
        report(f"Metric.dcg penalty {pen:g}", abs(Metric.dcg(gold["y"], gold["k_s"], penalty=pen) - float(gold[f"metric_dcg_pen/{pen:g}"])), 1e-9)
        report(f"Metric_for_Loss.dcg penalty {pen:g}", abs(float(Metric_for_Loss.dcg(y[4], 17, penalty=pen)) - float(gold[f"reward_dcg_pen/{pen:g}"][2, 16])), 1e-5)
    # loss + cut metrics in one pass (rlt_loss_metrics) == the separate kernels, bit for bit; at B = 5000 the grid strides
    # and above 16,384 lists the two-lists-per-wavefront pass strides too (odd B: its last wavefront works one half)
    for (Bf, Sf) in ((10, 300), (5000, 300), (9001, 100), (33, 777), (20001, 300)):
        g = torch.Generator().manual_seed(Bf)
        yy = (torch.rand(Bf, Sf, generator=g) < 0.2).float().to(dev)
        pp = torch.softmax(torch.randn(Bf, Sf, generator=g) * 2, 1).unsqueeze(2).to(dev)
        for cname in ("div_js_f1_aug1", "div_kl_dcg_aug0", "choopy_f1", "attncutloss_dcg"):
            crit = make_criterion(hl, cname)
            p1 = pp.clone().requires_grad_(True)
            l1 = crit(p1, yy)
            l1.backward()
            k1, f1, d1 = Metric.evaluate(p1, yy)
            p2 = pp.clone().requires_grad_(True)
            l2, k2, f2, d2 = Metric.step(crit, p2, yy)
            l2.backward()
            report(f"fused loss+metrics {cname} B{Bf} S{Sf}: loss", abs(float(l1) - float(l2)) / max(1.0, abs(float(l1))), 1e-6)
            report(f"fused loss+metrics {cname} B{Bf} S{Sf}: dp", float((p1.grad - p2.grad).abs().max()), 0)
            report(f"fused loss+metrics {cname} B{Bf} S{Sf}: k", float((k1 != k2).sum()), 0)
            report(f"fused loss+metrics {cname} B{Bf} S{Sf}: F1, DCG", max(abs(float(f1) - float(f2)), abs(float(d1) - float(d2)) / max(1.0, abs(float(d1)))), 1e-12)
    # list lengths around the round boundaries of the two-lists-per-wavefront pass (128 positions per round, S % 4 == 0,
    # S <= 384) and lengths only the general pass takes, against the oracle: loss, gradient and - fused - k, F1, DCG
    from oracle import metrics as om
    for S in (4, 40, 100, 128, 132, 200, 256, 260, 300, 301, 384, 388, 777):
        yy = (torch.rand(7, S) < 0.2).float()
        yy[3] = 0.0                                          # a list without relevant documents
        pp = torch.softmax(torch.randn(7, S) * 3, 1).unsqueeze(2)
        pp[5, :, 0] = 1.0 / S                                # all equal: the first position is the maximum
        for cname in ("div_js_f1_aug1", "div_kl_dcg_aug0", "choopy_dcg", "attncutloss_f1"):
            pr = pp.clone().requires_grad_(True)
            lr = make_criterion(ol, cname)(pr, yy)
            lr.backward()
            pg = pp.clone().to(dev).requires_grad_(True)
            lh, kh, f1h, dcgh = Metric.step(make_criterion(hl, cname), pg, yy.to(dev))
            lh.backward()
            report(f"loss {cname} S={S}", abs(lh.item() - lr.item()) / max(1, abs(lr.item())), 2e-5)
            report(f"dp   {cname} S={S}", rel(pg.grad, pr.grad), 1e-4)
            ko = om.cut_positions(pp.squeeze(2).numpy())
            report(f"k    {cname} S={S}", float((kh.cpu().numpy() != ko).sum()), 0)
            report(f"F1, DCG {cname} S={S}", max(abs(float(f1h) - om.Metric.f1(yy.numpy(), ko)),
                                                 abs(float(dcgh) - om.Metric.dcg(yy.numpy(), ko))), 1e-12)
    # cut probabilities underflowed into the denormal range: v_log_f32 / v_rcp_f32 read them as zero; the CE /
    # KL terms and the general pass scale them into the normal range first.  S = 300: two-lists-per-wavefront pass,
    # S = 301: general pass.  (The fast JS pass clamps at 2^-126 instead: its loss is exact, its gradient saturates.)
    for S in (300, 301):
        g = torch.Generator().manual_seed(S)
        yy = (torch.rand(6, S, generator=g) < 0.2).float()
        pp = torch.softmax(torch.randn(6, S, generator=g) * 2, 1)
        pp[:, 5::37] = torch.tensor([1e-39, 3e-41, 1e-44, 5e-40, 2e-38, 1e-42])[:, None]
        pp = pp.unsqueeze(2)
        for cname in ("attncutloss_f1", "div_kl_dcg_aug0", "div_kl_f1_aug1") + (("div_js_f1_aug1",) if S == 301 else ()):
            if cname.startswith("div_js"):
                # torch's own fp32 backward of the JS terms (x/y pieces of xlogy with a 3-bit denormal p) is off by 1e-2 from
                # the exact (ln p - ln m) / 2 at p = 1e-44; keep >= 13 significant bits so that the reference is a reference
                pp = pp.clone()
                pp[2, 5::37] = 4e-41
                pp[5, 5::37] = 7e-40
            pr = pp.clone().requires_grad_(True)
            lr = make_criterion(ol, cname)(pr, yy)
            lr.backward()
            pg = pp.clone().to(dev).requires_grad_(True)
            lh = make_criterion(hl, cname)(pg, yy.to(dev))
            lh.backward()
            report(f"denormal p: loss {cname} S={S}", abs(lh.item() - lr.item()) / max(1, abs(lr.item())), 2e-5)
            fin = torch.isfinite(pr.grad)
            # per entry: 1e-4 of its own size (the denormal positions carry gradients up to 1e36) or 1e-6 absolute
            report(f"denormal p: dp {cname} S={S} (finite entries)", float(((pg.grad.cpu() - pr.grad)[fin].abs() / (pr.grad[fin].abs() + 1e-2)).max()), 1e-4)
            report(f"denormal p: dp {cname} S={S} (same infinities)", float((torch.isfinite(pg.grad.cpu()) != fin).sum()), 0)
    # multi-task
    for tag, nt in (("t3", 3), ("t21", 2.1), ("t22", 2.2)):
        for metric in ("f1", "dcg"):
            lg = torch.from_numpy(gold["logits"]).to(dev)
            cl = torch.sigmoid(torch.from_numpy(gold["cls_logit"])).unsqueeze(2).to(dev).requires_grad_(True)
            rr = torch.from_numpy(gold["rerank"]).unsqueeze(2).to(dev).requires_grad_(True)
            p = torch.softmax(lg, 1).unsqueeze(2).requires_grad_(True)
            outs = [cl, rr, p] if nt == 3 else ([cl, p] if nt == 2.1 else [rr, p])
            loss = hl.MtCutLoss(metric=metric, rerank_weight=0.4, classi_weight=0.6, num_tasks=nt)(outs, y.to(dev))
            loss.backward()
            key = f"mtcut_{tag}_{metric}"
            report(f"mtcut loss {key}", abs(loss.item() - float(gold["loss/" + key])), 1e-5)
            if nt != 2.1:
                report(f"mtcut drerank {key}", float(np.abs(rr.grad.squeeze(2).cpu().numpy() - gold["drerank/" + key]).max()), 1e-7)
            if nt != 2.2:
                c = torch.sigmoid(torch.from_numpy(gold["cls_logit"]))
                dcl = torch.from_numpy(gold["dcls_logit/" + key]) / (c * (1 - c))     # d/dc from d/dlogit
                report(f"mtcut dclass {key}", rel(cl.grad.squeeze(2), dcl), 1e-4)


@section
def metrics():
    import golden_util as gu
    from utils.metrics import Metric
    gold = gu.load("losses_edge_s300")
    report("Metric.f1 golden", abs(Metric.f1(gold["y"], gold["k_s"]) - float(gold["metric_f1"])), 1e-6)
    report("Metric.dcg golden", abs(Metric.dcg(gold["y"], gold["k_s"]) - float(gold["metric_dcg"])), 1e-9)
    kx = np.array([[1, 0, 1], [0, 0, 1], [1, 0, 0]], dtype=np.float32)
    report("Metric.f1 KAT", abs(Metric.f1(kx, np.array([1, 2, 1])) - 0.5555555555555555), 1e-12)
    report("Metric.dcg KAT", abs(Metric.dcg(kx, np.array([1, 2, 1])) - 0.1230234154761809), 1e-12)
    tg = gu.load("task_metrics_s300")
    report("Metric.taskr_metric golden", abs(Metric.taskr_metric(tg["y"], tg["pred"]) - float(tg["taskr"])), 1e-9)
    report("Metric.taskc_metric golden", abs(Metric.taskc_metric(tg["y"], tg["pred"]) - float(tg["taskc"])), 1e-12)
    report("Metric.taskc_metric ties", abs(Metric.taskc_metric(tg["y"], tg["pred_ties"]) - float(tg["taskc_ties"])), 1e-12)
    p = torch.rand(33, 300)
    k, f1, dcg, _ = ops.cut_metrics(p.to(dev), torch.from_numpy(gold["y"][:1]).repeat(33, 1).to(dev))
    report("argmax", float((k.cpu().long() - (p.argmax(1) + 1)).abs().max()), 0)


@section
def layernorm():
    for (T, E) in [(1500, 256), (333, 128), (70, 64), (10, 512)]:
        x, r = torch.randn(T, E), torch.randn(T, E)
        g, b = torch.randn(E), torch.randn(E)
        dy = torch.randn(T, E)
        xr, rr, gr, br = [t.clone().double().requires_grad_(True) for t in (x, r, g, b)]
        yr = F.layer_norm(xr + rr, (E,), gr, br, 1e-5)
        yr.backward(dy.double())
        xd, rd, gd, bd = [t.clone().to(dev).requires_grad_(True) for t in (x, r, g, b)]
        yd = ops.add_layernorm(xd, rd, gd, bd)
        yd.backward(dy.to(dev))
        report(f"add_ln fwd {T}x{E}", rel(yd, yr), 5e-6)
        report(f"add_ln dx  {T}x{E}", rel(xd.grad, xr.grad), 2e-5)
        report(f"add_ln dr  {T}x{E}", rel(rd.grad, rr.grad), 2e-5)
        report(f"add_ln dg  {T}x{E}", rel(gd.grad, gr.grad), 2e-5)
        report(f"add_ln db  {T}x{E}", rel(bd.grad, br.grad), 2e-5)


def _pm(x):      # (B,S,E) -> (S*B,E)
    return x.permute(1, 0, 2).reshape(-1, x.shape[2]).contiguous()


def _unpm(x, B, S):
    return x.reshape(S, B, -1).permute(1, 0, 2).contiguous()


@section
def heads():
    for (B, S, E, kinds) in [(5, 300, 256, [0]), (7, 100, 128, [1, 2, 0]), (3, 40, 256, [1, 0])]:
        n = len(kinds)
        x = torch.randn(B, S, E)
        w, b = torch.randn(n, E) / 16, torch.randn(n)
        dout = torch.randn(n, B, S)
        xr, wr, br = [t.clone().double().requires_grad_(True) for t in (x, w, b)]
        outs = []
        for i, k in enumerate(kinds):
            z = xr @ wr[i] + br[i]
            outs.append(torch.softmax(z, 1) if k == 0 else (torch.sigmoid(z) if k == 1 else z))
        yr = torch.stack(outs)
        yr.backward(dout.double())
        xd = _pm(x).to(dev).requires_grad_(True)
        wd, bd = w.clone().to(dev).requires_grad_(True), b.clone().to(dev).requires_grad_(True)
        yd = ops.HeadsFn.apply(xd, wd, bd, kinds, S, B)
        yd.backward(dout.to(dev))
        report(f"heads fwd B{B} S{S} E{E} {kinds}", rel(yd, yr), 1e-5)
        report(f"heads dx  B{B} S{S} E{E} {kinds}", rel(_unpm(xd.grad, B, S), xr.grad), 2e-5)
        report(f"heads dw  B{B} S{S} E{E} {kinds}", rel(wd.grad, wr.grad), 2e-5)
        # the bias of a softmax head has an analytically zero gradient: compare absolutely
        report(f"heads db  B{B} S{S} E{E} {kinds}", float((bd.grad.double().cpu() - br.grad).abs().max()), 2e-5)


def _attn_ref(qkv, H):          # qkv (B,S,3E) double; attention over axis 0
    B, S, E3 = qkv.shape
    E = E3 // 3
    hd = E // H
    q, k, v = qkv.split(E, dim=2)
    sh = lambda t: t.reshape(B, S, H, hd).permute(1, 2, 0, 3)
    q, k, v = sh(q), sh(k), sh(v)
    sc = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    o = torch.softmax(sc, -1) @ v
    return o.permute(2, 0, 1, 3).reshape(B, S, E)


@section
def attention():
    for (B, S, H, HD) in [(5, 7, 4, 64), (63, 5, 4, 64), (130, 3, 2, 64), (200, 2, 4, 64), (5, 9, 8, 16), (70, 4, 8, 16),
                          (33, 3, 2, 32), (300, 2, 4, 64), (6, 3, 2, 128), (70, 2, 2, 128), (130, 2, 1, 128),
                          # whole 64-key tiles: the one-wavefront forward kernel of head dim 64 (1, 2, 5 tiles; 320 = a partial workgroup)
                          (64, 3, 4, 64), (128, 2, 2, 64), (320, 2, 4, 64),
                          # 512 lists and more in whole 64-row tiles: the pipelined forward of attention6h.hip (8 tiles; 9 tiles with a
                          # partly filled workgroup; 13 tiles, one head)
                          (512, 2, 4, 64), (576, 1, 2, 64), (832, 1, 1, 64)]:
        E = H * HD
        qkv = torch.randn(B, S, 3 * E)
        dout = torch.randn(B, S, E)
        qr = qkv.clone().double().requires_grad_(True)
        orf = _attn_ref(qr, H)
        orf.backward(dout.double())
        qd = _pm(qkv).to(dev).requires_grad_(True)
        od = ops.list_attention(qd, S, B, H)
        od.backward(_pm(dout).to(dev))
        report(f"attn fwd  B{B} S{S} H{H} HD{HD}", rel(_unpm(od, B, S), orf), mfma_tol(1e-5))
        g = _unpm(qd.grad, B, S)
        report(f"attn dq   B{B} S{S} H{H} HD{HD}", rel(g[..., :E], qr.grad[..., :E]), mfma_tol(3e-5))
        report(f"attn dk   B{B} S{S} H{H} HD{HD}", rel(g[..., E:2 * E], qr.grad[..., E:2 * E]), mfma_tol(3e-5))
        report(f"attn dv   B{B} S{S} H{H} HD{HD}", rel(g[..., 2 * E:], qr.grad[..., 2 * E:]), mfma_tol(3e-5))
    # sharp softmax (large scores) to exercise the online max
    B, S, H, HD = 150, 2, 1, 64
    qkv = torch.randn(B, S, 3 * HD) * 4
    qr = qkv.clone().double()
    od = ops.list_attention(_pm(qkv).to(dev), S, B, H)
    report("attn fwd sharp", rel(_unpm(od, B, S), _attn_ref(qr, H)), mfma_tol(2e-5))
    # head dim 16 around the tile (64), wavefront (32) and workgroup (128 / 256) boundaries of its kernels
    for B in (1, 17, 64, 65, 129, 257):
        S, H, HD = 2, 2, 16
        E = H * HD
        qkv = torch.randn(B, S, 3 * E)
        dout = torch.randn(B, S, E)
        qr = qkv.clone().double().requires_grad_(True)
        orf = _attn_ref(qr, H)
        orf.backward(dout.double())
        qd = _pm(qkv).to(dev).requires_grad_(True)
        od = ops.list_attention(qd, S, B, H)
        od.backward(_pm(dout).to(dev))
        report(f"attn hd16 boundaries B{B}: fwd", rel(_unpm(od, B, S), orf), mfma_tol(1e-5))
        report(f"attn hd16 boundaries B{B}: dqkv", rel(_unpm(qd.grad, B, S), qr.grad), mfma_tol(3e-5))
    # lazy rescaling of the forward kernels (the softmax reference moves only when a score exceeds it by 2^8 = 5.5 natural units):
    # scores that RISE by ~8.5 per 64-key tile (span 40) or ~60 (span 280) force the slow path in every tile, scores that FALL leave
    # the first tile's reference in place while the later weights underflow; a ramp of 6 stays inside the lazy window.  At span 280
    # one fp32 ulp of a score is 3e-5 and the weights of the top keys carry that as a RELATIVE error whatever the kernel does
    # (measured 2.5e-5 .. 6e-5 on the f32 MFMA): that case has the looser bound.
    for HD, B in ((16, 300), (64, 300), (64, 320), (64, 640)):      # (320: whole tiles; 640: the pipelined forward with its FIXED reference - the steep ramps raise its flags and take the fix-up launch)
        for tag, span, ftol in (("rising", 40.0, 2e-5), ("steeply rising", 280.0, 2e-4), ("falling", -280.0, 2e-5),
                                ("inside the window", 6.0, 2e-5)):
            S, H = 2, 2
            E = H * HD
            ramp = torch.linspace(0.0, 1.0, B).view(B, 1, 1)
            q = torch.full((B, S, E), 1.0) + 0.01 * torch.randn(B, S, E)
            k = ramp * (span / math.sqrt(HD)) * torch.ones(B, S, E) + 0.01 * torch.randn(B, S, E)
            v = torch.randn(B, S, E)
            qkv = torch.cat([q, k, v], dim=2)
            dout = torch.randn(B, S, E)
            qr = qkv.clone().double().requires_grad_(True)
            orf = _attn_ref(qr, H)
            orf.backward(dout.double())
            qd = _pm(qkv).to(dev).requires_grad_(True)
            od = ops.list_attention(qd, S, B, H)
            od.backward(_pm(dout).to(dev))
            report(f"attn lazy rescale hd{HD} B{B} scores {tag}: fwd", rel(_unpm(od, B, S), orf), mfma_tol(ftol))
            report(f"attn lazy rescale hd{HD} B{B} scores {tag}: dqkv", rel(_unpm(qd.grad, B, S), qr.grad), mfma_tol(5 * ftol))


@section
def lstm():
    import models as hm
    # I <= 3 takes the fused input projection (rlt_bilstm_rec_fwd_x) and rlt_narrow_dw; wider inputs the GEMM path
    for (B, S, I) in [(5, 12, 3), (40, 30, 3), (70, 9, 256), (33, 300, 3), (6, 7, 1), (6, 7, 2), (6, 7, 4)]:
        ref = torch.nn.LSTM(I, 128, num_layers=2, batch_first=True, bidirectional=True)
        x = torch.randn(B, S, I)
        dh = torch.randn(B, S, 256)
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)[0]
        yr.backward(dh)
        from models._common import ParamTree, bilstm
        pt = ParamTree(ref).to(dev)
        xd = _pm(x).to(dev).requires_grad_(True)
        yd = bilstm(xd, pt, S, B)
        yd.backward(_pm(dh).to(dev))
        report(f"bilstm fwd B{B} S{S} I{I}", rel(_unpm(yd, B, S), yr), 2e-5)
        report(f"bilstm dx  B{B} S{S} I{I}", rel(_unpm(xd.grad, B, S), xr.grad), 1e-4)
        for name, prm in ref.named_parameters():
            report(f"bilstm d{name} B{B} S{S}", rel(getattr(pt, name).grad, prm.grad), 1e-4)


@section
def lstm_generic():
    """Hidden sizes other than 128 (MMOECut's `encoding_size`, models/MMOECut.py:57,63): the general step-by-step form
    against nn.LSTM, forward and backward."""
    from models._common import ParamTree, bilstm
    for (B, S, I, Hd) in [(5, 12, 3, 64), (33, 20, 7, 96), (8, 300, 3, 64), (3, 1, 3, 32)]:
        torch.manual_seed(B + Hd)
        ref = torch.nn.LSTM(I, Hd, num_layers=2, batch_first=True, bidirectional=True)
        x, dh = torch.randn(B, S, I), torch.randn(B, S, 2 * Hd)
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)[0]
        yr.backward(dh)
        pt = ParamTree(ref).to(dev)
        xd = _pm(x).to(dev).requires_grad_(True)
        yd = bilstm(xd, pt, S, B)
        yd.backward(_pm(dh).to(dev))
        report(f"bilstm generic fwd B{B} S{S} I{I} H{Hd}", rel(_unpm(yd, B, S), yr), 2e-5)
        report(f"bilstm generic dx  B{B} S{S} I{I} H{Hd}", rel(_unpm(xd.grad, B, S), xr.grad), 1e-4)
        for name, prm in ref.named_parameters():
            report(f"bilstm generic d{name} B{B} S{S} H{Hd}", rel(getattr(pt, name).grad, prm.grad), 2e-4)
    # a whole model on it: MMOECut(encoding_size=64, d_model=128) against the oracle
    import models as hm
    from oracle import losses as ol, metrics as omet, models as om
    from oracle.weights import fill_state_dict, synthetic_lists
    from utils import losses as hl
    from utils.metrics import Metric
    kw = dict(seq_len=40, num_experts=3, num_tasks=3, encoding_size=64, d_model=128, n_head=4, dropout=0.0)
    refm = om.MMOECut(**kw)
    fill_state_dict(refm, 64)
    hipm = hm.MMOECut(**kw)
    hipm.load_state_dict(refm.state_dict())
    hipm = hipm.to(dev)
    x, y = synthetic_lists(6, 40, 3, 65)
    refm.train(), hipm.train()
    out_r, out_h = refm(x), hipm(x.to(dev))
    crit_r = ol.MtCutLoss(metric="f1", rerank_weight=0.4, classi_weight=0.6, num_tasks=3)
    crit_h = hl.MtCutLoss(metric="f1", rerank_weight=0.4, classi_weight=0.6, num_tasks=3)
    lr_, lh_ = crit_r(out_r, y), crit_h(out_h, y.to(dev))
    lr_.backward(), lh_.backward()
    for i, (a, b) in enumerate(zip(out_h, out_r)):
        report(f"mmoecut encoding_size=64 out{i}", float((a.detach().cpu() - b.detach()).abs().max()) / max(1.0, float(b.detach().abs().max())), 1e-5)
    report("mmoecut encoding_size=64 loss", abs(float(lh_) - float(lr_)), 1e-5)
    k_h, _f1, _dcg = Metric.evaluate(out_h[-1], y.to(dev))
    report("mmoecut encoding_size=64 k mismatches", float((k_h.cpu().numpy() != omet.cut_positions(out_r[-1].detach().squeeze(2).numpy())).sum()), 0)
    errs = _param_rel_l2(hipm, refm)
    worst = max(errs, key=errs.get)
    report(f"mmoecut encoding_size=64 grad rel-L2 worst ({worst})", errs[worst], 2e-3)


@section
def embed_mmoe():
    B, S = 6, 40
    score = torch.randn(B, S, 1)
    pe = torch.randn(S, 127)
    dout = torch.randn(B, S, 128)
    per = pe.clone().requires_grad_(True)
    outr = torch.cat((score, per.expand(B, S, 127)), 2)
    outr.backward(dout)
    ped = pe.clone().to(dev).requires_grad_(True)
    outd = ops.choopy_embed(score.to(dev), ped)
    outd.backward(_pm(dout).to(dev))
    report("choopy_embed fwd", rel(_unpm(outd, B, S), outr), 0)
    report("choopy_embed dpe", rel(ped.grad, per.grad), 1e-5)
    # gates + mix
    for (B, S, C, nt, ne) in [(5, 40, 256, 3, 3), (9, 300, 256, 2, 4)]:
        h = torch.randn(B, S, C)
        wg = [torch.randn(S * C, ne) / math.sqrt(S * C) for _ in range(nt)]
        ex = [torch.randn(B, S, C) for _ in range(ne)]
        dm = torch.randn(nt, B, S, C)
        hr = h.clone().double().requires_grad_(True)
        wr = [w.clone().double().requires_grad_(True) for w in wg]
        er = [e.clone().double().requires_grad_(True) for e in ex]
        gr = [torch.softmax(hr.reshape(B, -1) @ w, 1) for w in wr]
        est = torch.stack(er)
        mr = torch.stack([(g.t()[:, :, None, None] * est).sum(0) for g in gr])
        mr.backward(dm.double())
        hd = _pm(h).to(dev).requires_grad_(True)
        wd = [w.clone().to(dev).requires_grad_(True) for w in wg]
        ed = [_pm(e).to(dev).requires_grad_(True) for e in ex]
        gd = ops.MMOEGateFn.apply(hd, S, B, *wd)
        md = ops.MMOEMixFn.apply(gd, S, B, *ed)
        md.backward(torch.stack([_pm(dm[t]) for t in range(nt)]).to(dev))
        report(f"mmoe gates B{B} nt{nt} ne{ne}", rel(gd, torch.stack(gr)), 2e-5)
        report(f"mmoe mixed B{B}", rel(torch.stack([_unpm(md[t], B, S) for t in range(nt)]), mr), 2e-5)
        report(f"mmoe dh    B{B}", rel(_unpm(hd.grad, B, S), hr.grad), 1e-4)
        for t in range(nt):
            report(f"mmoe dwg{t} B{B}", rel(wd[t].grad, wr[t].grad), 1e-4)
        for e in range(ne):
            report(f"mmoe dex{e} B{B}", rel(_unpm(ed[e].grad, B, S), er[e].grad), 1e-4)


@section
def optimizer_and_trainer():
    """FusedAdam == torch.optim.Adam (coupled L2) after real training steps; run.py Trainer end to end."""
    import tempfile
    import models as hm
    from oracle import losses as olosses, models as omodels
    from oracle.weights import fill_state_dict, synthetic_lists
    from utils import losses as hl
    from rlt_hip.parallel import FlatModel, FusedAdam
    x, y = synthetic_lists(6, 300, 3, 11)
    ref = omodels.AttnCut(dropout=0.0)
    fill_state_dict(ref, 21)
    hip = hm.AttnCut(dropout=0.0)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(dev)
    flat = FlatModel(hip)
    opt_h = FusedAdam(flat, lr=1e-3, weight_decay=0.01)
    opt_r = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=0.01)
    crit_r = olosses.DivLoss(metric='f1', div_type='js')
    crit_h = hl.DivLoss(metric='f1', div_type='js')
    # both optimisers are driven by the SAME gradients (the HIP model's), so this isolates the Adam arithmetic:
    # Adam normalises by sqrt(v), so analytically-zero gradients (rounding noise) would otherwise move parameters
    # by +-lr with an implementation-dependent sign.
    for _ in range(3):
        opt_h.zero_grad()
        crit_h(hip(x.to(dev)), y.to(dev)).backward()
        for (n, a), (_, b) in zip(hip.named_parameters(), ref.named_parameters()):
            b.grad = a.grad.detach().cpu().clone()
        opt_r.step()
        opt_h.step()
    worst = 0.0
    for (n, a), (_, b) in zip(hip.named_parameters(), ref.named_parameters()):
        worst = max(worst, float((a.detach().cpu() - b.detach()).abs().max()))
    report("3 Adam steps on identical gradients: max |param diff| vs torch.optim.Adam", worst, 2e-6)
    # and the end-to-end loss after those steps agrees with the oracle evaluated at the oracle's parameters
    l_h = float(crit_h(hip(x.to(dev)), y.to(dev)))
    l_r = float(crit_r(ref(x), y))
    report("loss after 3 steps vs oracle", abs(l_h - l_r), 1e-4)
    # Trainer on a synthetic robust04-format set
    import run as hip_run
    with tempfile.TemporaryDirectory() as tmp:
        from dataloader import write_synthetic_robust04
        write_synthetic_robust04(tmp, "robust04", "drmm_tks", n_train=24, n_test=8, seq_len=300, seed=5)
        for name, extra in (("attncut", []), ("mmoecut", ["--num-tasks", "2.1", "--num-experts", "4"]),
                            ("choopy", []), ("mtattncut", [])):
            argv = ["--model-name", name, "--dataset-base", tmp, "--epochs", "2", "--use-conf", "0", "--batch-size", "8",
                    "--criterion", "f1", "--dropout", "0.1", "--lr", "1e-3", "--weight-decay", "0", "--seed", "3",
                    "--model-persist", "1", "--save-path", os.path.join(tmp, "ckpt"),
                    "--tensorboard-dir", os.path.join(tmp, "tb")] + extra
            f1, dcg = hip_run.main(argv)
            report(f"Trainer {name}: best test F1 in [0,1]", 0.0 if 0.0 <= f1 <= 1.0 and math.isfinite(dcg) else 1.0, 0)
            sd = torch.load(os.path.join(tmp, "ckpt", f"{name}.pkl"))
            kw = {"attncut": {}, "mmoecut": {"num_tasks": 2.1, "num_experts": 4}, "choopy": {}, "mtattncut": {}}[name]
            om = {"attncut": omodels.AttnCut, "mmoecut": omodels.MMOECut, "choopy": omodels.Choopy,
                  "mtattncut": omodels.MtAttnCut}[name](**kw)
            om.load_state_dict(sd)       # checkpoint loads into the reference-shaped module
            report(f"Trainer {name}: checkpoint loads into the oracle module", 0.0, 0)


@section
def dropout():
    """Dropout sites against torch references that use the SAME keep-masks (exported by the library)."""
    p, seed = 0.3, 12345
    # mask statistics
    m = torch.empty(4096, 256, device=dev)
    N.call("rlt_dropout_mask", seed, 4096, 256, p, N.ptr(m), N.stream())
    keep = float((m > 0).float().mean())
    report("dropout keep fraction", abs(keep - (1 - p)), 5e-3)
    report("dropout mask mean", abs(float(m.mean()) - 1.0), 1e-2)
    m2 = torch.empty_like(m)
    N.call("rlt_dropout_mask", seed + 1, 4096, 256, p, N.ptr(m2), N.stream())
    report("dropout masks decorrelated", abs(float(((m > 0) & (m2 > 0)).float().mean()) - (1 - p) ** 2), 5e-3)
    # residual + LayerNorm with dropout on the branch
    T, E = 700, 256
    x, r, g, b, dy = torch.randn(T, E), torch.randn(T, E), torch.randn(E), torch.randn(E), torch.randn(T, E)
    mk = torch.empty(T, E, device=dev)
    N.call("rlt_dropout_mask", seed, T, E, p, N.ptr(mk), N.stream())
    mkc = mk.cpu().double()
    xr, rr, gr, br = [t.clone().double().requires_grad_(True) for t in (x, r, g, b)]
    yr = F.layer_norm(xr + rr * mkc, (E,), gr, br, 1e-5)
    yr.backward(dy.double())
    xd, rd, gd, bd = [t.clone().to(dev).requires_grad_(True) for t in (x, r, g, b)]
    yd = ops.AddLayerNormFn.apply(xd, rd, gd, bd, 1e-5, p, seed)
    yd.backward(dy.to(dev))
    report("drop add_ln fwd", rel(yd, yr), 5e-6)
    report("drop add_ln dx", rel(xd.grad, xr.grad), 2e-5)
    report("drop add_ln dr", rel(rd.grad, rr.grad), 2e-5)
    report("drop add_ln dgamma", rel(gd.grad, gr.grad), 2e-5)
    # FFN with hidden dropout
    T, E, Fh = 300, 128, 512
    x = torch.randn(T, E)
    w1, b1, w2, b2 = torch.randn(Fh, E) / 11, torch.randn(Fh) / 10, torch.randn(E, Fh) / 22, torch.randn(E) / 10
    dy = torch.randn(T, E)
    mk = torch.empty(T, Fh, device=dev)
    N.call("rlt_dropout_mask", seed, T, Fh, p, N.ptr(mk), N.stream())
    mkc = mk.cpu().double()
    refs = [t.clone().double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    devs = [t.clone().to(dev).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    # ReLU is discontinuous at 0: a pre-activation within rounding distance of zero (|z| < 1e-4, about one element
    # per run at these sizes) legitimately takes either branch, and the branch decides an O(1) entry of dH.  On those
    # knife-edge elements the reference follows the branch the device took; everywhere else it is torch.relu.
    z = refs[0] @ refs[1].t() + refs[2]
    hd = torch.empty(T, Fh, device=dev)
    ops.gemm(0, 1, T, Fh, E, devs[0].detach(), E, devs[1].detach(), E, hd, Fh, bias=devs[2].detach(), flags=N.GEMM_RELU)
    gate = torch.where(z.detach().abs() < 1e-4, hd.cpu() > 0, z.detach() > 0).double()
    yr = (z * gate * mkc) @ refs[3].t() + refs[4]
    yr.backward(dy.double())
    yd = ops.FFNFn.apply(*devs, p, seed)
    yd.backward(dy.to(dev))
    report("drop ffn fwd", rel(yd, yr), mfma_tol(1e-5))
    for nm, a, bref in zip(("dx", "dw1", "db1", "dw2", "db2"), devs, refs):
        report(f"drop ffn {nm}", rel(a.grad, bref.grad), mfma_tol(3e-5))
    # attention-probability dropout
    for (B, S, H, HD) in [(70, 3, 2, 64), (40, 2, 4, 16), (192, 2, 2, 64)]:      # (192: whole tiles - the one-wavefront forward)
        E = H * HD
        qkv = torch.randn(B, S, 3 * E)
        dout = torch.randn(B, S, E)
        mk = torch.empty(S, H, B, B, device=dev)
        N.call("rlt_attention_dropout_mask", seed, S, B, H, p, N.ptr(mk), N.stream())
        mkc = mk.cpu().double()
        qr = qkv.clone().double().requires_grad_(True)
        q, k, v = qr.split(E, dim=2)
        sh = lambda t: t.reshape(B, S, H, HD).permute(1, 2, 0, 3)
        sc = (sh(q) @ sh(k).transpose(-1, -2)) / math.sqrt(HD)
        orf = ((torch.softmax(sc, -1) * mkc) @ sh(v)).permute(2, 0, 1, 3).reshape(B, S, E)
        orf.backward(dout.double())
        qd = _pm(qkv).to(dev).requires_grad_(True)
        od = ops.ListAttentionFn.apply(qd, S, B, H, p, seed)
        od.backward(_pm(dout).to(dev))
        report(f"drop attn fwd B{B} HD{HD}", rel(_unpm(od, B, S), orf), mfma_tol(1e-5))
        report(f"drop attn dqkv B{B} HD{HD}", rel(_unpm(qd.grad, B, S), qr.grad), mfma_tol(3e-5))
    # model level: train() with dropout runs, stays a distribution, differs from eval(); eval() == dropout 0
    import models as hm
    from oracle.weights import fill_state_dict, synthetic_lists
    from utils import losses as hl
    xm, ym = synthetic_lists(6, 300, 3, 5)
    a = hm.AttnCut(dropout=0.2)
    fill_state_dict(a, 3)
    a = a.to(dev)
    torch.manual_seed(0)
    a.train()
    pt = a(xm.to(dev))
    loss = hl.DivLoss(metric='f1', div_type='js')(pt, ym.to(dev))
    loss.backward()
    report("drop model rows sum to 1", float((pt.squeeze(2).sum(1) - 1).abs().max()), 1e-5)
    report("drop model grads finite", 0.0 if all(torch.isfinite(q.grad).all() for q in a.parameters()) else 1.0, 0)
    a.eval()
    pe = a(xm.to(dev))
    b0 = hm.AttnCut(dropout=0.0)
    fill_state_dict(b0, 3)
    b0 = b0.to(dev)
    report("eval() ignores dropout", float((pe - b0(xm.to(dev))).abs().max()), 0)
    report("train() dropout changes output", 0.0 if float((pe - pt).abs().max()) > 1e-7 else 1.0, 0)
    torch.manual_seed(0)
    ops._SEED_COUNTER[0] = 0
    a.train()
    p1 = a(xm.to(dev))
    torch.manual_seed(0)
    ops._SEED_COUNTER[0] = 0
    p2 = a(xm.to(dev))
    report("dropout reproducible under manual_seed", float((p1 - p2).abs().max()), 0)


@section
def bicut():
    """BiCut + BiCutLoss + the cut rule of run.py:131-136 against the reference goldens and the oracle."""
    import golden_util as gu
    import models as hm
    from oracle.cases import BICUT_CASES
    from oracle.weights import fill_state_dict, synthetic_lists
    from oracle import models as om
    from utils import losses as hl
    from utils.metrics import Metric
    for case in BICUT_CASES:
        gold = gu.load(case["tag"])
        model = hm.BiCut(dropout=0.0, **case["kwargs"])
        ref = om.BiCut(dropout=0.0, **case["kwargs"])
        fill_state_dict(ref, case["seed"])
        model.load_state_dict(ref.state_dict())
        model = model.to(dev)
        x, y = synthetic_lists(case["batch"], case["seq_len"], case["n_feat"], case["seed"] + 1)
        model.train()
        out = model(x.to(dev))
        report(f"{case['tag']} out", float(np.abs(out.detach().cpu().numpy() - gold["out0"]).max()), 1e-5)
        k, f1, _dcg = Metric.evaluate(out, y.to(dev))
        report(f"{case['tag']} k mismatches", float((k.cpu().numpy() != gold["k_s"]).sum()), 0)
        report(f"{case['tag']} f1", abs(float(f1) - float(gold["f1"])), 1e-4)
        for metric in case["criteria"]:
            o = model(x.to(dev))
            o.retain_grad()
            loss = hl.BiCutLoss(metric=metric)(o, y.to(dev))
            model.zero_grad()
            loss.backward()
            refl = float(gold["loss/" + metric])
            report(f"{case['tag']} loss {metric}", abs(loss.item() - refl) / max(1.0, abs(refl)), 1e-5)
            report(f"{case['tag']} dout {metric}", float(np.abs(o.grad.cpu().numpy() - gold["dout/" + metric]).max()
                                                         / max(1e-30, np.abs(gold["dout/" + metric]).max())), 1e-6)
            if metric == case["grad_crit"]:
                gu.check_grads_report(model, gold, report, case["tag"]) if hasattr(gu, "check_grads_report") else \
                    report(f"{case['tag']} grads", _grad_err(model, gold), 1e-3)
    gold = gu.load("bicutloss_edge_s50")
    y = torch.from_numpy(gold["y"]).to(dev)
    for metric in ("nci", "f1"):
        o = torch.softmax(torch.from_numpy(gold["logits"]), dim=2).to(dev).requires_grad_(True)
        loss = hl.BiCutLoss(metric=metric)(o, y)
        loss.backward()
        refl = float(gold["loss/" + metric])
        report(f"bicutloss edge loss {metric}", abs(loss.item() - refl) / max(1.0, abs(refl)), 1e-5)
        report(f"bicutloss edge dout {metric}", float(np.abs(o.grad.cpu().numpy() - gold["dout/" + metric]).max()), 1e-6)
    k, _f1, _dcg = Metric.evaluate(torch.softmax(torch.from_numpy(gold["logits"]), dim=2).to(dev), y)
    report("bicut cut rule edge rows", float((k.cpu().numpy() != gold["k_s"]).sum()), 0)
    # two-class head with dropout: rows still sum to 1, backward matches autograd on the same masks
    z = torch.randn(6 * 9, 2)
    zd = z.clone().to(dev).requires_grad_(True)
    od = ops.PairSoftmaxFn.apply(zd, 9, 6, 0.0, 0)
    g = torch.randn(6, 9, 2)
    od.backward(g.to(dev))
    zr = z.clone().double().requires_grad_(True)
    orf = torch.softmax(zr, 1).reshape(9, 6, 2).permute(1, 0, 2)
    orf.backward(g.double())
    report("pair_softmax fwd", rel(od, orf), 1e-6)
    report("pair_softmax bwd", rel(zd.grad, zr.grad), 1e-5)
    od = ops.PairSoftmaxFn.apply(zd, 9, 6, 0.3, 1234)
    report("pair_softmax dropout rows sum to 1", float((od.sum(2) - 1).abs().max()), 1e-6)


def _grad_err(model, gold):
    """max over parameters of |grad summary - golden| relative to the golden scale (norm / sum / probes)."""
    import golden_util as gu
    worst = 0.0
    for name, prm in model.named_parameters():
        flat = prm.grad.detach().reshape(-1).double().cpu()
        scale = max(float(gold["gnorm/" + name]), 1e-3)
        worst = max(worst, abs(float(flat.norm()) - float(gold["gnorm/" + name])) / scale)
        idx = torch.from_numpy(gu.probe_index(flat.numel(), name))
        worst = max(worst, float((flat[idx].numpy() - gold["gprobe/" + name]).__abs__().max()) / scale)
    return worst


@section
def models():
    import golden_util as gu
    import models as hm
    from oracle.cases import MODEL_CASES, make_criterion
    from utils import losses as hl
    from utils.metrics import Metric
    only = os.environ.get("PROBE_CASES")
    for case in MODEL_CASES:
        if only and case["tag"] not in only.split(","):
            continue
        try:
            gold = gu.load(case["tag"])
            t0 = time.time()
            model, x, y = gu.build(hm, case, device=dev)
            model.train()
            outs = gu.as_list(model(x))
            for i, o in enumerate(outs):
                # 1e-5 of the output's scale (probabilities: absolute; unnormalised rerank scores: relative);
                # north_star asks for 1e-4
                scale = max(1.0, float(np.abs(gold[f'out{i}']).max()))
                report(f"{case['tag']} out{i}", float(np.abs(o.detach().squeeze(2).cpu().numpy() - gold[f'out{i}']).max()) / scale, 1e-5)
            k, f1, dcg = Metric.evaluate(outs[-1], y)
            report(f"{case['tag']} k mismatches", float((k.cpu().numpy() != gold["k_s"]).sum()), 0)
            report(f"{case['tag']} f1", abs(float(f1) - float(gold["f1"])), 1e-4)
            report(f"{case['tag']} dcg", abs(float(dcg) - float(gold["dcg"])), 1e-4)
            for cname in case["criteria"]:
                loss = make_criterion(hl, cname, case)(model(x), y)
                ref = float(gold["loss/" + cname])
                report(f"{case['tag']} loss {cname}", abs(loss.item() - ref) / max(1, abs(ref)), 1e-4)
                if cname == case["grad_crit"]:
                    model.zero_grad()
                    loss.backward()
                    try:
                        worst = gu.check_grads(model, gold, rtol=1e-3, atol_frac=1e-3)
                        report(f"{case['tag']} grads", worst, 1e-3)
                    except AssertionError as exc:
                        print("   grad mismatch:", str(exc)[:300])
                        report(f"{case['tag']} grads", float("nan"), 1e-3)
            print(f"   ({case['tag']} took {time.time() - t0:.1f}s)", flush=True)
        except Exception:
            traceback.print_exc()
            report(f"{case['tag']} EXCEPTION", float("nan"), 0)



# ------------------------------------------------------------------------------------------------
# Benchmark-scale parity: the big-grid code paths (many row tiles per workgroup column,
# XCD block remap, split-K slabs over thousands of tiles, 1.2 M-row GEMMs) against independent references.
def _param_rel_l2(hip, ref):
    """Per-parameter relative L2 error of the gradients: |g_hip - g_ref| / max(|g_ref|, 1e-3 * largest |g_ref|)."""
    refs = dict(ref.named_parameters())
    top = max(float(v.grad.norm()) for v in refs.values() if v.grad is not None)
    out = {}
    for n, a in hip.named_parameters():
        g = refs[n].grad if refs[n].grad is not None else torch.zeros_like(refs[n])    # parameter not on the loss's path
        g = g.detach().cpu()                         # (the oracle module may have run in fp64 on the device)
        mine = a.grad.detach().cpu().double() if a.grad is not None else torch.zeros_like(g).double()
        out[n] = float((mine - g.double()).norm()) / max(float(g.norm()), 1e-3 * top)
    return out


SCALE_MODEL_CASES = [
    # tag, oracle/HIP class, kwargs, lists, positions, features, criterion
    ("attncut_b4096_s16", "AttnCut", {}, 4096, 16, 3, "div_js_f1_aug1"),
    ("attncut_b512_s300", "AttnCut", {}, 512, 300, 3, "div_js_f1_aug1"),
    ("choopy_b8192_s8", "Choopy", {"seq_len": 8}, 8192, 8, 1, "choopy_f1"),
    ("mmoecut_e4_t21_b1024_s40", "MMOECut", {"seq_len": 40, "num_experts": 4, "num_tasks": 2.1}, 1024, 40, 3, "mtcut_f1"),
]


@section
def scale_models():
    """Whole models at benchmark batch sizes against the CPU oracle (stock torch-CPU modules): outputs 1e-4 (the bound
    BASELINE.json states; observed ~1e-6), identical cut positions except lists whose two best positions are closer
    than 4e-6 in the oracle's own output (reported), loss 1e-4, and per-parameter gradient relative L2 within
    1e-3 (bf16x3) / 2e-4 (exact fp32) of the oracle's - about 10x what is observed - with the oracle's encoder layers
    following the device's ReLU branch on the (counted) knife-edge units (flip_aligned_grads explains why); in the default bf16x6
    mode the un-aligned figure is asserted too (worst 1.2e-3, median 1.5e-4: 10x the observed values)."""
    import models as hm
    from oracle import losses as ol, metrics as omet, models as om
    from oracle.cases import make_criterion
    from oracle.weights import fill_state_dict, synthetic_lists
    from utils import losses as hl
    from utils.metrics import Metric
    only = os.environ.get("PROBE_CASES")
    bf = N.get_precision() == "bf16x3"
    grad_tol = 1e-3 if bf else 2e-4
    edge = 3e-4 if bf else 4e-6
    for tag, cls, kw, B, S, F_, cname in SCALE_MODEL_CASES:
        if only and tag not in only.split(","):
            continue
        t0 = time.time()
        case = {"kwargs": kw, "w_r": 0.4, "w_c": 0.6}
        ref = getattr(om, cls)(dropout=0.0, **kw)
        fill_state_dict(ref, 900 + B % 97)
        hip = getattr(hm, cls)(dropout=0.0, **kw)
        hip.load_state_dict(ref.state_dict())
        hip = hip.to(dev)
        x, y = synthetic_lists(B, S, F_, 901 + S)
        ref.train(), hip.train()
        out_h = hip(x.to(dev))
        outs_h = list(out_h) if isinstance(out_h, (list, tuple)) else [out_h]
        nodes = _encoder_nodes(outs_h)                   # the device's ReLU decisions, before backward frees the stash
        loss_h = make_criterion(hl, cname, case)(out_h, y.to(dev))
        loss_h.backward()
        # un-aligned oracle first: its outputs, loss and cut positions are THE reference; its gradients the printed
        # un-aligned figure, asserted with a bound of its own in the default bf16x6 mode (the second oracle backward doubles the CPU
        # time of the section: the other two modes go without)
        plain = None
        if N.get_precision() == "bf16x6":
            out_r = ref(x)
            loss_r = make_criterion(ol, cname, case)(out_r, y)
            loss_r.backward()
            plain = _param_rel_l2(hip, ref)
            ref.zero_grad()
        else:
            with torch.no_grad():
                out_r = ref(x)
                loss_r = make_criterion(ol, cname, case)(out_r, y)
        outs_r = list(out_r) if isinstance(out_r, (list, tuple)) else [out_r]
        outs_r = [o.detach() for o in outs_r]
        counts = _align_relu(ref, hip, nodes, B, S, edge)
        make_criterion(ol, cname, case)(ref(x), y).backward()
        t_cpu = time.time() - t0
        for i, (a, b) in enumerate(zip(outs_h, outs_r)):
            scale = max(1.0, float(b.abs().max()))
            report(f"{tag} out{i} max|d|", float((a.detach().cpu() - b).abs().max()) / scale, 1e-4)
            # relative to the values themselves (p ~ 1/S): the bound a wrong tile would break by orders of magnitude
            report(f"{tag} out{i} rel", rel(a, b), 2e-4)
        p_r = outs_r[-1].squeeze(2)
        k_r = omet.cut_positions(p_r.numpy())
        k_h, f1_h, dcg_h = Metric.evaluate(outs_h[-1], y.to(dev))
        k_h = k_h.cpu().numpy()
        top2 = torch.topk(p_r, 2, dim=1).values
        gap = (top2[:, 0] - top2[:, 1]).numpy()
        differ = k_h != k_r
        report(f"{tag} k mismatches outside knife-edge lists (gap >= 4e-6)", float((differ & (gap >= 4e-6)).sum()), 0)
        report(f"{tag} k mismatches on knife-edge lists (of {int((gap < 4e-6).sum())})", float(differ.sum()), float((gap < 4e-6).sum()))
        report(f"{tag} F1", abs(float(f1_h) - omet.Metric.f1(y.numpy(), k_h)), 1e-6)
        report(f"{tag} DCG", abs(float(dcg_h) - omet.Metric.dcg(y.numpy(), k_h)), 1e-6)
        report(f"{tag} F1 vs oracle k", abs(float(f1_h) - omet.Metric.f1(y.numpy(), k_r)), 1e-4)
        report(f"{tag} loss", abs(float(loss_h) - float(loss_r)) / max(1.0, abs(float(loss_r))), 1e-4)
        report(f"{tag} ReLU decisions differing outside |z| < {edge:g}", float(counts[1]), 0)
        errs = _param_rel_l2(hip, ref)
        worst = max(errs, key=errs.get)
        report(f"{tag} grad rel-L2 worst, flip-aligned ({worst})", errs[worst], grad_tol)
        med = sorted(errs.values())[len(errs) // 2]
        report(f"{tag} grad rel-L2 median, flip-aligned", med, grad_tol / 5)
        diag = ""
        if plain is not None:
            wp = max(plain, key=plain.get)
            pmed = sorted(plain.values())[len(plain) // 2]
            diag = f"un-aligned worst rel-L2 {plain[wp]:.2e} ({wp}), median {pmed:.2e}; "
            # against the oracle's OWN ReLU decisions (no alignment): the knife-edge units add their whole gradient to the difference,
            # so the bound is ~10x what round 5 observed (worst 1.1e-4 - the narrow bias_ih_l0 of the BiLSTM -, median up to 1.5e-5)
            report(f"{tag} grad rel-L2 worst, un-aligned ({wp})", plain[wp], 1.2e-3)
            report(f"{tag} grad rel-L2 median, un-aligned", pmed, 1.5e-4)
        print(f"   ({tag}: {counts[0]} of {counts[2]} FFN units aligned; {diag}oracle {t_cpu:.1f}s, total {time.time() - t0:.1f}s)", flush=True)
        del hip, out_h, loss_h, nodes
        torch.cuda.empty_cache()


def _chunks(n, step):
    for lo in range(0, n, step):
        yield lo, min(n, lo + step)


@section
@section
def scale_attention(shapes=((4096, 2, 4, 64), (8192, 2, 8, 16))):
    """List-axis attention at the batch sizes of the BASELINE configurations against fp64 (every element compared): 16 / 32 query + key
    tiles per column."""
    for (B, S, H, HD) in shapes:
        E = H * HD
        g = torch.Generator(device="cpu").manual_seed(B + HD)
        qkv = torch.randn(B, S, 3 * E, generator=g)
        dout = torch.randn(B, S, E, generator=g)
        qr = qkv.to(dev).double().requires_grad_(True)
        orf = _attn_ref(qr, H)
        orf.backward(dout.to(dev).double())
        qd = _pm(qkv).to(dev).requires_grad_(True)
        od = ops.list_attention(qd, S, B, H)
        od.backward(_pm(dout).to(dev))
        report(f"attn fwd  B{B} S{S} H{H} HD{HD}", rel(_unpm(od, B, S), orf), mfma_tol(1e-5))
        gq = _unpm(qd.grad, B, S)
        report(f"attn dq   B{B} S{S} H{H} HD{HD}", rel(gq[..., :E], qr.grad[..., :E]), mfma_tol(3e-5))
        report(f"attn dk   B{B} S{S} H{H} HD{HD}", rel(gq[..., E:2 * E], qr.grad[..., E:2 * E]), mfma_tol(3e-5))
        report(f"attn dv   B{B} S{S} H{H} HD{HD}", rel(gq[..., 2 * E:], qr.grad[..., 2 * E:]), mfma_tol(3e-5))
        del qr, orf, qd, od, gq
        torch.cuda.empty_cache()


@section
def scale_ops():
    """Kernel families at the shapes of the 4096 x 300 step against fp64 references (every element compared)."""
    scale_attention()
    # GEMMs with 1,228,800 rows (= 4096 lists x 300 positions): all four layouts
    T = 4096 * 300
    g = torch.Generator(device=dev).manual_seed(5)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    for (Nn, K) in [(768, 256), (256, 2048), (128, 256)]:
        A, W, bias = rn(T, K), rn(Nn, K) / math.sqrt(K), rn(Nn)
        C = torch.empty(T, Nn, device=dev)
        ops.gemm(0, 1, T, Nn, K, A, K, W, K, C, Nn, bias=bias)
        worst, top = 0.0, 0.0
        for lo, hi in _chunks(T, 131072):
            ref = A[lo:hi].double() @ W.double().t() + bias.double()
            worst = max(worst, float((C[lo:hi].double() - ref).abs().max()))
            top = max(top, float(ref.abs().max()))
        report(f"gemm NT {T}x{Nn}x{K} (all rows)", worst / top, mfma_tol(2e-6 * math.sqrt(K) + 1e-6))
    for (Nn, K) in [(256, 768), (256, 2048)]:
        A, Wm, C0 = rn(T, K), rn(K, Nn) / math.sqrt(K), rn(T, Nn)
        C = C0.clone()
        ops.gemm(0, 0, T, Nn, K, A, K, Wm, Nn, C, Nn, flags=N.GEMM_ACCUMULATE)
        worst, top = 0.0, 0.0
        for lo, hi in _chunks(T, 131072):
            ref = A[lo:hi].double() @ Wm.double() + C0[lo:hi].double()
            worst = max(worst, float((C[lo:hi].double() - ref).abs().max()))
            top = max(top, float(ref.abs().max()))
        report(f"gemm NN accumulate {T}x{Nn}x{K} (all rows)", worst / top, mfma_tol(2e-6 * math.sqrt(K) + 1e-6))
    # dW products: K = 1,228,800 (split-K slabs pinned to XCDs) with the bias gradient riding on them
    for (M, Nn) in [(2048, 256), (768, 256), (256, 2048), (512, 128)]:
        A, Bm = rn(T, M), rn(T, Nn)
        C = torch.empty(M, Nn, device=dev)
        cs = torch.empty(M, device=dev)
        ops.gemm(1, 0, M, Nn, T, A, M, Bm, Nn, C, Nn, colsum_a=cs)
        ref = torch.zeros(M, Nn, dtype=torch.float64, device=dev)
        csr = torch.zeros(M, dtype=torch.float64, device=dev)
        for lo, hi in _chunks(T, 131072):
            ad = A[lo:hi].double()
            ref += ad.t() @ Bm[lo:hi].double()
            csr += ad.sum(0)
        report(f"gemm TN split-K {M}x{Nn}x{T}", rel(C, ref), mfma_tol(2e-6 * math.sqrt(300) + 1e-6))
        report(f"gemm TN colsum  {M}x{Nn}x{T}", rel(cs, csr), 2e-5)
        del A, Bm
    A, Bm = rn(40000, 256), rn(256, 40000)
    C = torch.empty(256, 256, device=dev)
    ops.gemm(1, 1, 256, 256, 40000, A, 256, Bm, 40000, C, 256)
    report("gemm TT 256x256x40000", rel(C, A.double().t() @ Bm.double().t()), mfma_tol(2e-6 * math.sqrt(300) + 1e-6))
    # FFN pair with the 1-bit mask at full height
    Fh, E = 2048, 256
    A, W, bias = rn(T, E), rn(Fh, E) / 16, rn(Fh) / 10
    Hh = torch.empty(T, Fh, device=dev)
    bits = ops.alloc_relu_bits(T, Fh, dev)
    ops.gemm_bits(0, 1, T, Fh, E, A, E, W, E, Hh, Fh, bias=bias, flags=N.GEMM_RELU, bits_out=bits)
    worst, top, badbits = 0.0, 0.0, 0
    for lo, hi in _chunks(T, 65536):
        ref = torch.relu(A[lo:hi].double() @ W.double().t() + bias.double())
        worst = max(worst, float((Hh[lo:hi].double() - ref).abs().max()))
        top = max(top, float(ref.abs().max()))
        un = ops.unpack_relu_bits(bits[lo // 32:hi // 32], hi - lo)
        badbits += int((un != (Hh[lo:hi] > 0)).sum())
    report(f"gemm_bits relu fwd {T}x{Fh}x{E} (all rows)", worst / top, mfma_tol(1e-5))
    report(f"gemm_bits mask bits {T}x{Fh}", float(badbits), 0)
    dY, W2 = rn(T, E), rn(E, Fh) / 16
    dH = torch.empty(T, Fh, device=dev)
    ops.gemm_bits(0, 0, T, Fh, E, dY, E, W2, Fh, dH, Fh, bits_in=bits, mask_scale=1.0)
    worst, top = 0.0, 0.0
    for lo, hi in _chunks(T, 65536):
        ref = (dY[lo:hi].double() @ W2.double()) * (Hh[lo:hi] > 0)
        worst = max(worst, float((dH[lo:hi].double() - ref).abs().max()))
        top = max(top, float(ref.abs().max()))
    report(f"gemm_bits masked bwd {T}x{Fh}x{E} (all rows)", worst / top, mfma_tol(2e-6 * 16 + 1e-6))
    del A, W, Hh, bits, dY, W2, dH
    torch.cuda.empty_cache()
    # BiLSTM: 128 workgroups per direction
    from models._common import ParamTree, bilstm
    for (B, S, I) in [(4096, 12, 3), (4100, 5, 256)]:
        torch.manual_seed(B)
        ref = torch.nn.LSTM(I, 128, num_layers=2, batch_first=True, bidirectional=True)
        x, dh = torch.randn(B, S, I), torch.randn(B, S, 256)
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)[0]
        yr.backward(dh)
        pt = ParamTree(ref).to(dev)
        xd = _pm(x).to(dev).requires_grad_(True)
        yd = bilstm(xd, pt, S, B)
        yd.backward(_pm(dh).to(dev))
        report(f"bilstm fwd B{B} S{S} I{I}", rel(_unpm(yd, B, S), yr), 2e-5)
        report(f"bilstm dx  B{B} S{S} I{I}", rel(_unpm(xd.grad, B, S), xr.grad), 1e-4)
        for name, prm in ref.named_parameters():
            report(f"bilstm d{name} B{B} S{S}", rel(getattr(pt, name).grad, prm.grad), 2e-4)
    # residual + LayerNorm and the heads at full height (HBM-streaming kernels; every row compared)
    x, r, gmm, bt, dy = rn(T, 256), rn(T, 256), rn(256), rn(256), rn(T, 256)
    xd, rd, gd, bd = [t.clone().requires_grad_(True) for t in (x, r, gmm, bt)]
    yd = ops.add_layernorm(xd, rd, gd, bd)
    yd.backward(dy)
    worst = 0.0
    dgr = torch.zeros(256, dtype=torch.float64, device=dev)
    wdx = 0.0
    for lo, hi in _chunks(T, 131072):
        xr = (x[lo:hi].double() + r[lo:hi].double()).requires_grad_(True)
        gr = gmm.double().requires_grad_(True)
        yr = F.layer_norm(xr, (256,), gr, bt.double(), 1e-5)
        yr.backward(dy[lo:hi].double())
        worst = max(worst, float((yd[lo:hi].double() - yr).abs().max()))
        wdx = max(wdx, float((xd.grad[lo:hi].double() - xr.grad).abs().max()))
        dgr += gr.grad
    report(f"add_ln fwd {T}x256 (all rows)", worst, 2e-5)
    report(f"add_ln dx  {T}x256 (all rows)", wdx, 1e-4)
    report(f"add_ln dgamma {T}x256", rel(gd.grad, dgr), 5e-5)


def _attn_mask_pairs(seed, pair0, npair, B, p):
    """keep-masks (npair,B,B) of the attention-probability dropout for pairs pair0.. (pair = position * H + head)."""
    mk = torch.empty(npair, B, B, device=dev)
    N.call("rlt_attention_dropout_mask_range", seed, pair0, npair, B, p, N.ptr(mk), N.stream())
    return mk


@section
def scale_dropout():
    """The TRAIN-MODE (dropout > 0) kernel variants at the shapes of the benchmark steps: they are
    separate instantiations (`attn3_fwd/bwd_dkv/bwd_dq_kernel<.,true>` with their per-tile double-buffered hash tables and
    the compiler-scheduled tile body, `attn_*_kernel<.,.,true>` in exact-fp32 mode, the dropout epilogue of `rlt_gemm_bits`,
    `add_ln_*` with the branch dropout).  The reference trains with these rates (hyper_parameter_drmm_tks.conf:41: 0.4;
    models/Choopy.py:7: 0.2).  References: fp64 on the device with the kernels' own keep-masks exported as data."""
    # ---- list-axis attention with probability dropout: 16 / 32 key tiles per column, forward + dQ / dK / dV
    for (B, S, H, HD, p) in [(4096, 2, 4, 64, 0.4), (8192, 2, 8, 16, 0.2)]:
        E = H * HD
        seed = 77000 + B
        g = torch.Generator(device="cpu").manual_seed(B + HD + 1)
        qkv = torch.randn(B, S, 3 * E, generator=g)
        dout = torch.randn(B, S, E, generator=g)
        qd = _pm(qkv).to(dev).requires_grad_(True)
        od = ops.ListAttentionFn.apply(qd, S, B, H, p, seed)
        od.backward(_pm(dout).to(dev))
        o_dev, g_dev = _unpm(od.detach(), B, S), _unpm(qd.grad, B, S)
        # fp64 reference one (position, head) at a time: a 4096^2 / 8192^2 score matrix and its mask per iteration
        q64 = qkv.to(dev).double()
        d64 = dout.to(dev).double()
        o_ref = torch.empty(B, S, E, dtype=torch.float64, device=dev)
        g_ref = torch.empty(B, S, 3 * E, dtype=torch.float64, device=dev)
        keep_sum = 0.0
        for s_ in range(S):
            for h in range(H):
                mk = _attn_mask_pairs(seed, s_ * H + h, 1, B, p)[0].double()
                keep_sum += float((mk > 0).double().mean())
                sl = lambda part: slice(part * E + h * HD, part * E + (h + 1) * HD)
                q, k, v = [q64[:, s_, sl(i)].clone().requires_grad_(True) for i in range(3)]
                o = ((torch.softmax(q @ k.t() / math.sqrt(HD), -1) * mk) @ v)
                o.backward(d64[:, s_, h * HD:(h + 1) * HD])
                o_ref[:, s_, h * HD:(h + 1) * HD] = o.detach()
                for i, t in enumerate((q, k, v)):
                    g_ref[:, s_, sl(i)] = t.grad
                del mk, q, k, v, o
        report(f"drop attn keep fraction B{B} HD{HD} p{p}", abs(keep_sum / (S * H) - (1 - p)), 2e-3)
        report(f"drop attn fwd  B{B} S{S} H{H} HD{HD} p{p}", rel(o_dev, o_ref), mfma_tol(1e-5))
        report(f"drop attn dq   B{B} S{S} H{H} HD{HD} p{p}", rel(g_dev[..., :E], g_ref[..., :E]), mfma_tol(3e-5))
        report(f"drop attn dk   B{B} S{S} H{H} HD{HD} p{p}", rel(g_dev[..., E:2 * E], g_ref[..., E:2 * E]), mfma_tol(3e-5))
        report(f"drop attn dv   B{B} S{S} H{H} HD{HD} p{p}", rel(g_dev[..., 2 * E:], g_ref[..., 2 * E:]), mfma_tol(3e-5))
        del qd, od, q64, d64, o_ref, g_ref, o_dev, g_dev
        torch.cuda.empty_cache()
    # ---- FFN hidden dropout fused into the rlt_gemm_bits epilogue at 1,228,800 rows, and the masked dH product
    T, Fh, E, p, seed = 4096 * 300, 2048, 256, 0.4, 424243
    gg = torch.Generator(device=dev).manual_seed(6)
    rn = lambda *s_: torch.randn(*s_, device=dev, generator=gg)
    A, W, bias = rn(T, E), rn(Fh, E) / 16, rn(Fh) / 10
    Hd = torch.empty(T, Fh, device=dev)
    bits = ops.alloc_relu_bits(T, Fh, dev)
    ops.gemm_bits(0, 1, T, Fh, E, A, E, W, E, Hd, Fh, bias=bias, flags=N.GEMM_RELU, bits_out=bits, drop_p=p, seed=seed)
    mk = torch.empty(T, Fh, device=dev)
    N.call("rlt_dropout_mask", seed, T, Fh, p, N.ptr(mk), N.stream())
    worst, top, badbits, near = 0.0, 0.0, 0, 0
    for lo, hi in _chunks(T, 65536):
        z = A[lo:hi].double() @ W.double().t() + bias.double()
        ref = torch.relu(z) * mk[lo:hi].double()
        worst = max(worst, float((Hd[lo:hi].double() - ref).abs().max()))
        top = max(top, float(ref.abs().max()))
        un = ops.unpack_relu_bits(bits[lo // 32:hi // 32], hi - lo)
        badbits += int((un != (Hd[lo:hi] > 0)).sum())
        # the bit must be "passed the ReLU and kept": compare with the fp64 decision away from the knife edge
        want = (z > 0) & (mk[lo:hi] > 0)
        near += int(((un != want) & (z.abs() > 1e-4)).sum())
    report(f"gemm_bits relu+dropout fwd {T}x{Fh}x{E} p{p} (all rows)", worst / top, mfma_tol(1e-5))
    report(f"gemm_bits relu+dropout bits == (output > 0) {T}x{Fh}", float(badbits), 0)
    report(f"gemm_bits relu+dropout bits == fp64 (z > 0 and kept) outside |z| < 1e-4", float(near), 0)
    report(f"gemm_bits dropout keep fraction {T}x{Fh}", abs(float((mk > 0).float().mean()) - (1 - p)), 1e-3)
    dY, W2 = rn(T, E), rn(E, Fh) / 16
    dH = torch.empty(T, Fh, device=dev)
    ops.gemm_bits(0, 0, T, Fh, E, dY, E, W2, Fh, dH, Fh, bits_in=bits, mask_scale=1.0 / (1.0 - p))
    worst, top = 0.0, 0.0
    for lo, hi in _chunks(T, 65536):
        ref = (dY[lo:hi].double() @ W2.double()) * (Hd[lo:hi] > 0) / (1.0 - p)
        worst = max(worst, float((dH[lo:hi].double() - ref).abs().max()))
        top = max(top, float(ref.abs().max()))
    report(f"gemm_bits masked+scaled bwd {T}x{Fh}x{E} p{p} (all rows)", worst / top, mfma_tol(2e-6 * 16 + 1e-6))
    del A, W, Hd, bits, mk, dY, W2, dH
    torch.cuda.empty_cache()
    # ---- residual + LayerNorm with the branch dropout at 1,228,800 rows, forward and backward (every row compared)
    p, seed = 0.4, 515151
    x, r, gmm, bt, dy = rn(T, 256), rn(T, 256), rn(256), rn(256), rn(T, 256)
    xd, rd, gd, bd = [t.clone().requires_grad_(True) for t in (x, r, gmm, bt)]
    yd = ops.AddLayerNormFn.apply(xd, rd, gd, bd, 1e-5, p, seed)
    yd.backward(dy)
    mk = torch.empty(T, 256, device=dev)
    N.call("rlt_dropout_mask", seed, T, 256, p, N.ptr(mk), N.stream())
    worst = wdx = wdr = 0.0
    dgr = torch.zeros(256, dtype=torch.float64, device=dev)
    dbr = torch.zeros(256, dtype=torch.float64, device=dev)
    for lo, hi in _chunks(T, 131072):
        xr = x[lo:hi].double().requires_grad_(True)
        rr = r[lo:hi].double().requires_grad_(True)
        gr, br = gmm.double().requires_grad_(True), bt.double().requires_grad_(True)
        yr = F.layer_norm(xr + rr * mk[lo:hi].double(), (256,), gr, br, 1e-5)
        yr.backward(dy[lo:hi].double())
        worst = max(worst, float((yd[lo:hi].double() - yr).abs().max()))
        wdx = max(wdx, float((xd.grad[lo:hi].double() - xr.grad).abs().max()))
        wdr = max(wdr, float((rd.grad[lo:hi].double() - rr.grad).abs().max()))
        dgr += gr.grad
        dbr += br.grad
    report(f"drop add_ln fwd {T}x256 p{p} (all rows)", worst, 2e-5)
    report(f"drop add_ln dx  {T}x256 p{p} (all rows)", wdx, 1e-4)
    report(f"drop add_ln dr  {T}x256 p{p} (all rows)", wdr, 2e-4)
    report(f"drop add_ln dgamma {T}x256 p{p}", rel(gd.grad, dgr), 5e-5)
    report(f"drop add_ln dbeta  {T}x256 p{p}", rel(bd.grad, dbr), 5e-5)
    del x, r, dy, xd, rd, yd, mk
    torch.cuda.empty_cache()
    # ---- whole encoder layers at FULL size in train mode: position-subset fp64 reference with all four masks exported
    _full_encoder_cases([("attncut 4096x300 E256 H4 dropout 0.4", 4096, 256, 4, None, 0.4, [0, 149, 299]),
                         ("choopy 8192x300 E128 H8 dropout 0.2", 8192, 128, 8, None, 0.2, [1, 298])], 300)



def _saved_hidden(y):
    """The FFN hidden activation (T, 2048) the encoder-layer tape node of `y` keeps for its backward."""
    return ops.encoder_ffn_hidden(y)


@section
def full_size_kernels():
    """BASELINE configs[1] (AttnCut 4096 x 300) and configs[2] (Choopy 8192 x 300) at FULL size: sparse-cotangent
    checks of the two big kernel chains against independent references.  Lists are independent in the BiLSTM and
    positions are independent in the encoder layer, so a reference on a SUBSET of lists / positions pins the
    corresponding rows of the full-size forward, and - with the upstream gradient zero outside the subset - every
    weight gradient and the subset's input-gradient rows of the full-size backward (all other rows must be exactly
    zero).  The kernels still run their full grids."""
    import bench
    from models._common import ParamTree, bilstm
    from oracle import explicit
    only = os.environ.get("PROBE_CASES", "")
    # ---- (1a) BiLSTM 4096 x 300, list subset vs nn.LSTM on the CPU
    B, S = 4096, 300
    x, _y = bench.synth_batch(B, S, 3, 77, "cpu")
    torch.manual_seed(3)
    ref = torch.nn.LSTM(3, 128, num_layers=2, batch_first=True, bidirectional=True)
    sub = torch.tensor([0, 1, 31, 32, 33, 63, 64, 1000, 2047, 2048, 3333, 4064, 4094, 4095])
    dh_sub = torch.randn(len(sub), S, 256)
    xr = x[sub].clone().requires_grad_(True)
    yr = ref(xr)[0]
    yr.backward(dh_sub)
    pt = ParamTree(ref).to(dev)
    xd = _pm(x).to(dev).requires_grad_(True)
    yd = bilstm(xd, pt, S, B)
    dh = torch.zeros(B, S, 256)
    dh[sub] = dh_sub
    yd.backward(_pm(dh).to(dev))
    yh = yd.detach().reshape(S, B, 256)[:, sub.to(dev)].permute(1, 0, 2)
    report(f"full bilstm fwd {B}x{S}: {len(sub)} lists vs nn.LSTM", rel(yh, yr), 2e-5)
    gx = xd.grad.reshape(S, B, 3)
    report("full bilstm dx subset rows", rel(gx[:, sub.to(dev)].permute(1, 0, 2), xr.grad), 1e-4)
    mask = torch.ones(B, dtype=torch.bool, device=dev)
    mask[sub.to(dev)] = False
    report("full bilstm dx rows outside the subset are zero", float(gx[:, mask].abs().max()), 0)
    for name, prm in ref.named_parameters():
        report(f"full bilstm d{name}", rel(getattr(pt, name).grad, prm.grad), 2e-4)
    h_full = yd.detach()
    del xd, yd, dh
    torch.cuda.empty_cache()
    # ---- (1b) encoder layers at full size, position subset vs the fp64 restatement
    cases = [("attncut 4096x300 E256 H4", 4096, 256, 4, h_full, 0.0, [0, 1, 149, 299]),
             ("choopy 8192x300 E128 H8", 8192, 128, 8, None, 0.0, [0, 1, 149, 299])]
    _full_encoder_cases([c for c in cases if not only or c[0].split()[0] in only], S)
    del h_full


def _full_encoder_cases(cases, S):
    """Encoder layers at full size against the fp64 restatement (oracle/explicit.py, run on the device) on a SUBSET of
    positions with a sparse upstream gradient; `p_drop > 0`: train mode, the reference takes the layer's four keep-masks
    (attention probabilities, both residual branches, FFN hidden) as data from the library's mask exports."""
    from models._common import ParamTree
    from oracle import explicit
    for (tag, Bq, E, H, src, p_drop, pos) in cases:
        T = S * Bq
        torch.manual_seed(11)
        layer = torch.nn.TransformerEncoderLayer(d_model=E, nhead=H, dropout=p_drop)
        FF = layer.linear1.out_features
        pl = ParamTree(layer).to(dev)
        if src is None:
            src = torch.randn(T, E, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        xin = src.clone().requires_grad_(True)
        torch.manual_seed(12)
        y = ops.encoder_layer(xin, pl, S, Bq, H, p_drop)
        rows = torch.cat([torch.arange(s * Bq, (s + 1) * Bq, device=dev) for s in pos])
        hid_dev = _saved_hidden(y)[rows] > 0                       # the device's ReLU (and keep) decisions on the subset rows
        seeds = y.grad_fn.cfg[7]
        gsub = torch.randn(len(rows), E, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
        dy = torch.zeros(T, E, device=dev)
        dy[rows] = gsub
        y.backward(dy)
        # reference: (B, |pos|, E) fp64 through oracle/explicit.py on the device (torch fp64 ops, independent of librlt).
        # ReLU units whose fp64 pre-activation is within rounding distance of zero follow the device's branch
        # (counted below): a flipped unit legitimately moves a row of dX and of linear1's gradient by O(1e-3).
        edge = 2e-4 if N.get_precision() == "bf16x3" else 2e-6
        to_bse = lambda t: t.reshape(len(pos), Bq, -1).permute(1, 0, 2)
        gate_dev = to_bse(hid_dev)
        masks, kept = None, None
        if p_drop > 0:
            s_attn, s_ln1, s_ffn, s_ln2 = seeds

            def row_mask(seed, cols):
                mk = torch.empty(T, cols, device=dev)
                N.call("rlt_dropout_mask", seed, T, cols, p_drop, N.ptr(mk), N.stream())
                return to_bse(mk[rows]).double()
            masks = {"attn": torch.stack([_attn_mask_pairs(s_attn, s * H, H, Bq, p_drop) for s in pos]).double(),
                     "res1": row_mask(s_ln1, E), "ffn": row_mask(s_ffn, FF), "res2": row_mask(s_ln2, E)}
            kept = masks["ffn"] > 0
            for name, mk in masks.items():
                report(f"full encoder {tag}: keep fraction of the {name} mask", abs(float((mk > 0).double().mean()) - (1 - p_drop)), 3e-3)
        flips = [0, 0]

        def gate(z):
            own = z > 0
            near = z.abs() < edge
            dev_dec = gate_dev if kept is None else torch.where(kept, gate_dev, own)    # a dropped unit shows no decision
            flips[0] += int((near & (own != dev_dec)).sum())
            flips[1] += int((~near & (own != dev_dec)).sum())
            return torch.where(near, dev_dec, own).to(z.dtype)

        xs = to_bse(src[rows]).double().requires_grad_(True)
        prm64 = {k: v.detach().to(dev).double().requires_grad_(True) for k, v in layer.state_dict().items()}
        yr = explicit.encoder_layer(xs, prm64, "", H, relu_gate=gate, masks=masks)
        yr.backward(to_bse(gsub).double())
        report(f"full encoder {tag}: ReLU decisions differing from fp64 outside |z| < {edge:g}", float(flips[1]), 0)
        print(f"   ({tag}: {flips[0]} knife-edge ReLU units of {hid_dev.numel()} follow the device's branch)", flush=True)
        report(f"full encoder fwd {tag}: positions {pos}", rel(to_bse(y.detach()[rows]), yr), mfma_tol(2e-5))
        report(f"full encoder dx {tag}", rel(to_bse(xin.grad[rows]), xs.grad), mfma_tol(1e-4))
        m = torch.ones(T, dtype=torch.bool, device=dev)
        m[rows] = False
        report(f"full encoder dx {tag}: rows outside the subset are zero", float(xin.grad[m].abs().max()), 0)
        for name, prm in pl.named_parameters():
            g64 = prm64[name].grad
            err = float((prm.grad.double() - g64).norm() / g64.norm())
            report(f"full encoder d{name} {tag}", err, mfma_tol(1e-4))
        del xin, y, dy, xs, yr, prm64, src, masks, kept
        torch.cuda.empty_cache()


@section
def full_size_models():
    """Whole training step of configs[1] and configs[2] at full size: exact-fp32 mode against the default bf16x3 mode on
    the same inputs and weights (a self-comparison: it comes in addition to scale_models / scale_ops /
    full_size_kernels, which compare with independent references)."""
    import bench
    import models as hm
    from utils import losses as hl
    from utils.metrics import Metric
    only = os.environ.get("PROBE_CASES", "")
    S = 300
    for (tag, mk, crit, Bq, F_) in [
            ("attncut_b4096_s300", lambda: hm.AttnCut(dropout=0.0), lambda: hl.DivLoss(metric='f1', div_type='js', augmented=True), 4096, 3),
            ("choopy_b8192_s300", lambda: hm.Choopy(dropout=0.0), lambda: hl.ChoopyLoss(metric='f1'), 8192, 1)]:
        if only and tag.split("_")[0] not in only:
            continue
        from oracle.weights import fill_state_dict
        xg, yg = bench.synth_batch(Bq, S, F_, 20240, dev)
        res = {}
        keep = N.get_precision()
        for mode in ("fp32", "bf16x3", "bf16x6"):
            N.set_precision(mode)
            model = mk()
            fill_state_dict(model, 55)
            model = model.to(dev)
            model.train()
            p = model(xg)
            loss = crit()(p, yg)
            loss.backward()
            k, f1, dcg = Metric.evaluate(p, yg)
            res[mode] = (p.detach().squeeze(2), float(loss), k.cpu(), float(f1), float(dcg),
                         {n: q.grad.detach().clone() for n, q in model.named_parameters()})
            del model, p, loss
            torch.cuda.empty_cache()
        N.set_precision(keep)
        pa, la, ka, fa, da, ga = res["fp32"]
        pb, lb, kb, fb, db, gb = res["bf16x3"]
        report(f"{tag} rows sum to 1 (fp32 mode)", float((pa.double().sum(1) - 1).abs().max()), 1e-5)
        report(f"{tag} p: fp32 mode vs bf16x3 max|d|", float((pa - pb).abs().max()), 1e-4)
        report(f"{tag} p: fp32 mode vs bf16x3 relative", rel(pb, pa), 1e-3)
        top2 = torch.topk(pa, 2, dim=1).values
        gap = (top2[:, 0] - top2[:, 1]).cpu()
        differ = ka != kb
        report(f"{tag} k differs between modes outside knife-edge lists", float((differ & (gap >= 4e-6)).sum()), 0)
        print(f"   ({tag}: {int(differ.sum())} of {Bq} cut positions differ between modes; {int((gap < 4e-6).sum())} knife-edge lists)")
        report(f"{tag} loss between modes", abs(la - lb) / max(1.0, abs(la)), 1e-4)
        report(f"{tag} F1 between modes", abs(fa - fb), 2e-3)
        worst = max(float((ga[n] - gb[n]).norm() / max(float(ga[n].norm()), 1e-30)) for n in ga if float(ga[n].norm()) > 1e-6)
        report(f"{tag} gradient rel-L2 between modes (worst parameter)", worst, 3e-2)
        # the fp32-faithful mode against the exact-fp32 mode at full size: both carry 24-bit products, so they agree at the
        # level of fp32 rounding (bounds ~10x what is observed), two orders tighter than bf16x3 does above
        pc, lc, kc, fc, dc_, gc = res["bf16x6"]
        report(f"{tag} p: fp32 mode vs bf16x6 max|d|", float((pa - pc).abs().max()), 4e-7)
        report(f"{tag} p: fp32 mode vs bf16x6 relative", rel(pc, pa), 2e-5)
        differ6 = ka != kc
        report(f"{tag} k differs between fp32 and bf16x6 outside knife-edge lists", float((differ6 & (gap >= 4e-6)).sum()), 0)
        print(f"   ({tag}: {int(differ6.sum())} of {Bq} cut positions differ between fp32 and bf16x6)")
        report(f"{tag} loss fp32 vs bf16x6", abs(la - lc) / max(1.0, abs(la)), 2e-6)
        worst6 = max(float((ga[n] - gc[n]).norm() / max(float(ga[n].norm()), 1e-30)) for n in ga if float(ga[n].norm()) > 1e-6)
        report(f"{tag} gradient rel-L2 fp32 vs bf16x6 (worst parameter, ReLU flips included)", worst6, 2e-4)
        del res, pa, pb, ga, gb, pc, gc
        torch.cuda.empty_cache()



@section
def full_size_oracle():
    """BASELINE configs[1] AT FULL SIZE against the oracle, directly ("F1@k vs CPU ref"): AttnCut forward on the
    4096 x 300 benchmark batch (bench.synth_batch, seed 20240) + DivLoss(js, f1, augmented) + the cut metrics, HIP (the
    process's precision mode; the test runs it in the default bf16x6) against the CPU oracle's forward pass on the same
    weights and inputs: max |dp| <= 1e-4 (the bound BASELINE.json states), identical cut positions outside the (counted)
    knife-edge lists, |dF1|, |dDCG|, |dloss| <= 1e-4.  The oracle's encoder layer runs over the position axis in chunks: its
    list-axis attention, FFN and norms treat the positions independently, so chunking changes nothing
    but the size of the 4096 x 4096 score matrices held at once."""
    import bench
    import models as hm
    from oracle import losses as ol, metrics as omet, models as om
    from oracle.weights import fill_state_dict
    from utils import losses as hl
    from utils.metrics import Metric
    Bq, S = 4096, 300
    t0 = time.time()
    xg, yg = bench.synth_batch(Bq, S, 3, 20240, dev)
    ref = om.AttnCut(dropout=0.0)
    fill_state_dict(ref, 55)
    hip = hm.AttnCut(dropout=0.0)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(dev)
    hip.train()
    with torch.no_grad():
        p_h = hip(xg)
        loss_h = hl.DivLoss(metric='f1', div_type='js', augmented=True)(p_h, yg)
        k_h, f1_h, dcg_h = Metric.evaluate(p_h, yg)
    torch.cuda.synchronize()
    t_hip = time.time() - t0
    x, y = xg.cpu(), yg.cpu()
    try:
        torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    except AttributeError:
        pass
    ref.train()
    t0 = time.time()
    with torch.no_grad():
        h = ref.encoding_layer(x)[0]
        enc = torch.cat([ref.attention_layer(h[:, s0:s0 + 12].contiguous()) for s0 in range(0, S, 12)], dim=1)
        p_r = ref.decison_layer(enc)
        loss_r = ol.DivLoss(metric='f1', div_type='js', augmented=True)(p_r, y)
    t_cpu = time.time() - t0
    p_r2 = p_r.squeeze(2)
    report("full_size_oracle attncut_b4096_s300 p max|d| (BASELINE bound 1e-4)", float((p_h.squeeze(2).cpu() - p_r2).abs().max()), 1e-4)
    report("full_size_oracle attncut_b4096_s300 p relative", rel(p_h.squeeze(2), p_r2), 2e-4)
    k_r = omet.cut_positions(p_r2.numpy())
    k_hn = k_h.cpu().numpy()
    top2 = torch.topk(p_r2, 2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]).numpy()
    differ = k_hn != k_r
    report("full_size_oracle attncut_b4096_s300 k mismatches outside knife-edge lists (gap >= 4e-6)", float((differ & (gap >= 4e-6)).sum()), 0)
    report(f"full_size_oracle attncut_b4096_s300 k mismatches on knife-edge lists (of {int((gap < 4e-6).sum())})", float(differ.sum()),
           float((gap < 4e-6).sum()))
    f1_r, dcg_r = omet.Metric.f1(y.numpy(), k_r), omet.Metric.dcg(y.numpy(), k_r)
    report("full_size_oracle attncut_b4096_s300 |dF1| vs the oracle's own cut positions", abs(float(f1_h) - f1_r), 1e-4)
    report("full_size_oracle attncut_b4096_s300 |dDCG| vs the oracle's own cut positions", abs(float(dcg_h) - dcg_r) / max(1.0, abs(dcg_r)), 1e-4)
    report("full_size_oracle attncut_b4096_s300 |dloss|", abs(float(loss_h) - float(loss_r)) / max(1.0, abs(float(loss_r))), 1e-4)
    print(f"   (full_size_oracle: HIP {t_hip:.1f} s incl. data, oracle forward {t_cpu:.1f} s on {torch.get_num_threads()} threads; "
          f"F1 {float(f1_h):.6f} / {f1_r:.6f}, DCG {float(dcg_h):.5f} / {dcg_r:.5f}, loss {float(loss_h):.6e} / {float(loss_r):.6e})", flush=True)
    del hip, p_h
    torch.cuda.empty_cache()


@section
def full_size_oracle_choopy():
    """BASELINE configs[2] AT FULL SIZE against the oracle, directly: Choopy forward on the 8192 x 300 batch (HIP,
    the process's precision mode) against the CPU oracle on a SUBSET of 24 positions of the same batch.  Choopy has no recurrence: the
    position-encoded rows of one position pass through the three encoder layers (list-axis attention, FFN, norms) and the decision
    Linear without meeting another position (reference models/Choopy.py:19-22) - only the final softmax over
    the 300 positions couples them, and it cancels in log p[:, s] - log p[:, s'] = z_s - z_s'.  So the oracle's logits on 24 positions
    (2 positions per chunk: two 8192 x 8192 score matrices per head at a time) pin the device's probabilities on those positions for
    all 8192 lists: | (log p_s - log p_s0) - (z_s - z_s0) | <= 1e-4 for every list and subset position."""
    import bench
    import models as hm
    from oracle import models as om
    from oracle.weights import fill_state_dict
    Bq, S = 8192, 300
    t0 = time.time()
    xg, _yg = bench.synth_batch(Bq, S, 1, 20241, dev)
    ref = om.Choopy(seq_len=S, dropout=0.0)
    fill_state_dict(ref, 56)
    hip = hm.Choopy(seq_len=S, dropout=0.0)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(dev)
    hip.train()
    with torch.no_grad():
        p_h = hip(xg).squeeze(2).double().cpu()
    torch.cuda.synchronize()
    t_hip = time.time() - t0
    del hip
    torch.cuda.empty_cache()
    try:
        torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    except AttributeError:
        pass
    pos = sorted(set([0, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 299] + list(range(17, 300, 29))))[:24]
    x = xg.cpu()
    ref.train()
    t0 = time.time()
    zs = []
    with torch.no_grad():
        for i in range(0, len(pos), 2):
            pp = pos[i:i + 2]
            h = torch.cat((x[:, pp], ref.position_encoding[pp].expand(Bq, len(pp), 127)), dim=2)
            h = ref.attention_layer(h)
            zs.append(ref.decison_layer[0](h).squeeze(2).double())       # the Linear of the softmax head: logits
    z = torch.cat(zs, dim=1)                                             # (B, 24)
    t_cpu = time.time() - t0
    lp = torch.log(p_h[:, pos])
    live = torch.isfinite(lp).all(dim=1)
    d = (lp - lp[:, :1]) - (z - z[:, :1])
    report(f"full_size_oracle_choopy b8192_s300: lists with finite log p on the {len(pos)} positions", float((~live).sum()), 0)
    report("full_size_oracle_choopy b8192_s300 max | (log p_s - log p_s0) - (z_s - z_s0) | over all lists", float(d[live].abs().max()), 1e-4)
    # the same on the probabilities renormalised over the subset (what a softmax over these 24 positions alone would give)
    q_h = p_h[:, pos] / p_h[:, pos].sum(1, keepdim=True)
    q_r = torch.softmax(z, dim=1)
    report("full_size_oracle_choopy b8192_s300 max |dq| on the subset-renormalised probabilities", float((q_h - q_r).abs().max()), 1e-4)
    report("full_size_oracle_choopy b8192_s300 arg max over the subset, lists that differ outside knife edges (gap >= 4e-6)",
           float(((q_h.argmax(1) != q_r.argmax(1)) & ((torch.topk(q_r, 2, dim=1).values[:, 0] - torch.topk(q_r, 2, dim=1).values[:, 1]) >= 4e-6)).sum()), 0)
    print(f"   (full_size_oracle_choopy: HIP {t_hip:.1f} s incl. data, oracle {len(pos)} positions {t_cpu:.1f} s on {torch.get_num_threads()} threads; "
          f"max |d log-ratio| {float(d[live].abs().max()):.2e})", flush=True)


# ------------------------------------------------------------------------------------------------
def _encoder_nodes(outs):
    """linear1.weight data_ptr -> the device's ReLU decisions (bool, on the host) of every encoder-layer tape node
    reachable from `outs`; call BEFORE backward (the tape frees its stash afterwards)."""
    found, seen, stack = {}, set(), [o.grad_fn for o in outs if o.grad_fn is not None]
    while stack:
        node = stack.pop()
        if node is None or node in seen:
            continue
        seen.add(node)              # the node itself (not its id): keeps the python wrapper alive, ids are not reused
        if type(node).__name__ == "EncoderLayerFnBackward":
            w1 = node.saved_tensors[7]                     # x, then the 12 weights in rlt_encoder_weights order
            assert w1.dim() == 2 and w1.shape[0] == 2048, tuple(w1.shape)
            found[w1.data_ptr()] = (ops.encoder_ffn_hidden(node) > 0).cpu()
        stack.extend(fn for fn, _ in node.next_functions)
    return found


def _align_relu(ref, hip, nodes, B, S, edge):
    """Make every nn.TransformerEncoderLayer of the oracle model `ref` follow the DEVICE's ReLU decision on units whose
    own pre-activation is within `edge` of zero.  Returns the counters [aligned knife-edge units, units that differ
    outside the edge (must stay 0), units in total]."""
    hip_params = dict(hip.named_parameters())
    counts = [0, 0, 0]
    n_layers = 0
    for name, mod in ref.named_modules():
        if not isinstance(mod, torch.nn.TransformerEncoderLayer):
            continue
        gate_dev = nodes[hip_params[name + ".linear1.weight"].data_ptr()].reshape(S, B, -1).permute(1, 0, 2)
        n_layers += 1

        def act(z, gate_dev=gate_dev):
            gate_dev = gate_dev.to(z.device)                 # (the oracle module may run in fp64 on the device)
            own = z.detach() > 0
            near = z.detach().abs() < edge
            counts[0] += int((near & (own != gate_dev)).sum())
            counts[1] += int((~near & (own != gate_dev)).sum())
            counts[2] += own.numel()
            return z * torch.where(near, gate_dev, own).to(z.dtype)
        mod.activation = act
    assert n_layers == len(nodes), (n_layers, len(nodes))
    return counts


@section
def scale_mmoe():
    """MMOECut (models/MMOECut.py:86-110) and the multi-task criterion at a C4-like per-GPU shape (VERDICT r02 items
    3b / weak 4): (i) the gate product (B, 76800) @ (76800, n_e), its softmax, the mixture and their gradients at
    B = 2048 x 300 against fp64 on the device; (ii) RerankLoss / BCE partial sums (rlt_mt_terms) and their gradients at
    4096 x 300 against the CPU oracle; (iii) the WHOLE model MMOECut(4 experts, tasks 2.1) at 1024 x 300 with MtCutLoss
    against the oracle MODULE evaluated in float64 on the device (same restatement as the CPU oracle, stock torch
    modules; the criterion runs on the host on its outputs) - outputs, cut positions, loss, flip-aligned gradients."""
    import models as hm
    from oracle import losses as ol, metrics as omet, models as om
    from oracle.weights import fill_state_dict, synthetic_lists
    from utils import losses as hl
    from utils.metrics import Metric
    # ---- (i) gates + mixture at B = 2048, S = 300, C = 256 (K = 76,800), 2 tasks x 4 experts
    B, S, C, nt, ne = 2048, 300, 256, 2, 4
    g = torch.Generator(device=dev).manual_seed(21)
    rn = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    h = rn(S * B, C)
    wg = [rn(S * C, ne) / math.sqrt(S * C) * 3 for _ in range(nt)]
    ex = [rn(S * B, C) for _ in range(ne)]
    dm = rn(nt, S * B, C)
    hd_ = h.clone().requires_grad_(True)
    wd = [w.clone().requires_grad_(True) for w in wg]
    ed = [e.clone().requires_grad_(True) for e in ex]
    gd = ops.MMOEGateFn.apply(hd_, S, B, *wd)
    md = ops.MMOEMixFn.apply(gd, S, B, *ed)
    md.backward(dm)
    h64 = _unpm(h, B, S).double().requires_grad_(True)                      # (B,S,C)
    w64 = [w.double().requires_grad_(True) for w in wg]
    e64 = [_unpm(e, B, S).double().requires_grad_(True) for e in ex]
    g64 = [torch.softmax(h64.reshape(B, -1) @ w, 1) for w in w64]
    m64 = torch.stack([sum(gt[:, e_, None, None] * e64[e_] for e_ in range(ne)) for gt in g64])       # (nt,B,S,C)
    m64.backward(torch.stack([_unpm(dm[t], B, S) for t in range(nt)]).double())
    report(f"mmoe gates B{B} S{S} K{S * C} nt{nt} ne{ne}", rel(gd, torch.stack(g64)), 2e-5)
    report(f"mmoe mixed B{B} S{S}", rel(torch.stack([_unpm(md[t].detach(), B, S) for t in range(nt)]), m64), 2e-5)
    report(f"mmoe dh    B{B} S{S}", rel(_unpm(hd_.grad, B, S), h64.grad), 1e-4)
    for t in range(nt):
        report(f"mmoe dwg{t} B{B} S{S}", rel(wd[t].grad, w64[t].grad), 1e-4)
    for e_ in range(ne):
        report(f"mmoe dex{e_} B{B} S{S}", rel(_unpm(ed[e_].grad, B, S), e64[e_].grad), 1e-4)
    del h, wg, ex, dm, hd_, wd, ed, gd, md, h64, w64, e64, g64, m64
    torch.cuda.empty_cache()
    # ---- (ii) multi-task terms at 4096 x 300: the three task codes of MtCutLoss against the CPU oracle
    B, S = 4096, 300
    x, y = synthetic_lists(B, S, 1, 4242)
    gcpu = torch.Generator().manual_seed(7)
    cut = torch.softmax(torch.randn(B, S, generator=gcpu) * 2, 1).unsqueeze(2)
    rer = (torch.randn(B, S, generator=gcpu) * 0.01 + 0.02 * y).unsqueeze(2)      # hinge active or not by the margin
    cls = torch.sigmoid(torch.randn(B, S, generator=gcpu)).unsqueeze(2)
    for nt_ in (3, 2.1, 2.2):
        for scale in (1.0, -1.0):                            # -1: relevant documents score LOWER -> the hinge is active
            def outs(device, dtype=torch.float32):
                c, r_, p_ = [t.clone().to(device).requires_grad_(True) for t in (cls, rer * scale, cut)]
                return [c, r_, p_] if nt_ == 3 else ([c, p_] if nt_ == 2.1 else [r_, p_])
            o_r, o_h = outs("cpu"), outs(dev)
            l_r = ol.MtCutLoss(metric="f1", rerank_weight=0.4, classi_weight=0.6, num_tasks=nt_)(o_r, y)
            l_h = hl.MtCutLoss(metric="f1", rerank_weight=0.4, classi_weight=0.6, num_tasks=nt_)(o_h, y.to(dev))
            l_r.backward(), l_h.backward()
            tag = f"mtcut {B}x{S} tasks {nt_:g} hinge {'active' if scale < 0 else 'inactive'}"
            report(f"{tag}: loss", abs(float(l_h) - float(l_r)) / max(1.0, abs(float(l_r))), 1e-5)
            for i, (a, b) in enumerate(zip(o_h, o_r)):
                gb = b.grad if b.grad is not None else torch.zeros_like(b)      # inactive hinge: the oracle returns a constant 0
                report(f"{tag}: d out{i}", float((a.grad.cpu() - gb).abs().max() / max(1e-30, float(gb.abs().max()))), 1e-4)
    # ---- (iii) the whole model at 1024 x 300 against the oracle module in float64 on the device
    B, S = 1024, 300
    kw = dict(seq_len=S, num_experts=4, num_tasks=2.1)
    bf = N.get_precision() == "bf16x3"
    edge, grad_tol = (3e-4, 1e-3) if bf else (4e-6, 2e-4)
    t0 = time.time()
    ref = om.MMOECut(dropout=0.0, **kw)
    fill_state_dict(ref, 977)
    hip = hm.MMOECut(dropout=0.0, **kw)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(dev)
    ref = ref.double().to(dev)
    x, y = synthetic_lists(B, S, 3, 978)
    ref.train(), hip.train()
    out_h = hip(x.to(dev))
    nodes = _encoder_nodes(list(out_h))
    crit_h = hl.MtCutLoss(metric="f1", rerank_weight=0.4, classi_weight=0.6, num_tasks=2.1)
    crit_r = ol.MtCutLoss(metric="f1", rerank_weight=0.4, classi_weight=0.6, num_tasks=2.1)
    loss_h = crit_h(out_h, y.to(dev))
    loss_h.backward()
    counts = _align_relu(ref, hip, nodes, B, S, edge)
    out_r = ref(x.to(dev).double())
    loss_r = crit_r([o.cpu().float() for o in out_r], y)          # the criterion in the oracle's own arithmetic (fp32, host)
    loss_r.backward()
    for i, (a, b) in enumerate(zip(out_h, out_r)):
        report(f"mmoecut_e4_t21_b{B}_s{S} out{i} max|d|", float((a.detach().double() - b.detach()).abs().max()), 1e-5)
        report(f"mmoecut_e4_t21_b{B}_s{S} out{i} rel", rel(a, b), 2e-4)
    p_r = out_r[-1].detach().squeeze(2).cpu()
    k_r = omet.cut_positions(p_r.float().numpy())
    k_h, f1_h, dcg_h = Metric.evaluate(out_h[-1], y.to(dev))
    top2 = torch.topk(p_r, 2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]).numpy()
    differ = k_h.cpu().numpy() != k_r
    report(f"mmoecut_e4_t21_b{B}_s{S} k mismatches outside knife-edge lists (gap >= 4e-6)", float((differ & (gap >= 4e-6)).sum()), 0)
    print(f"   ({int(differ.sum())} cut positions differ, {int((gap < 4e-6).sum())} knife-edge lists of {B}; "
          f"{len(set(k_r.tolist()))} distinct cut positions)", flush=True)
    report(f"mmoecut_e4_t21_b{B}_s{S} F1 vs oracle k", abs(float(f1_h) - omet.Metric.f1(y.numpy(), k_r)), 1e-4)
    report(f"mmoecut_e4_t21_b{B}_s{S} loss", abs(float(loss_h) - float(loss_r)) / max(1.0, abs(float(loss_r))), 1e-4)
    report(f"mmoecut_e4_t21_b{B}_s{S} ReLU decisions differing outside |z| < {edge:g}", float(counts[1]), 0)
    errs = _param_rel_l2(hip, ref)
    worst = max(errs, key=errs.get)
    report(f"mmoecut_e4_t21_b{B}_s{S} grad rel-L2 worst, flip-aligned ({worst})", errs[worst], grad_tol)
    report(f"mmoecut_e4_t21_b{B}_s{S} grad rel-L2 median, flip-aligned", sorted(errs.values())[len(errs) // 2], grad_tol / 5)
    print(f"   ({counts[0]} of {counts[2]} FFN units aligned; {time.time() - t0:.1f}s)", flush=True)


FLIP_CASES = [
    ("attncut_b33_s64", "AttnCut", {}, 33, 64, 3, "div_js_f1_aug1"),
    ("mtattncut_t3_b9_s33", "MtAttnCut", {"num_tasks": 3}, 9, 33, 3, "mtcut_f1"),
    ("choopy_b16_s40", "Choopy", {"seq_len": 40}, 16, 40, 1, "choopy_f1"),
    ("mmoecut_e3_t3_b7_s40", "MMOECut", {"seq_len": 40, "num_experts": 3, "num_tasks": 3}, 7, 40, 3, "mtcut_f1"),
    ("attncut_b300_s50", "AttnCut", {}, 300, 50, 3, "div_js_f1_aug1"),
]


@section
def flip_aligned_grads():
    """Whole-model gradients against the CPU oracle with the ReLU knife edge taken out: the oracle's encoder layers
    follow the device's branch on the (counted) FFN units whose pre-activation is within rounding distance of zero.
    What remains must agree to 1e-4 (exact-fp32 mode) / 1e-3 (bf16x3) of each parameter's gradient norm - this turns
    DESIGN.md section 2's explanation of the looser un-aligned figures into a test."""
    import models as hm
    from oracle import losses as ol, models as om
    from oracle.cases import make_criterion
    from oracle.weights import fill_state_dict, synthetic_lists
    from utils import losses as hl
    bf = N.get_precision() == "bf16x3"
    edge, tol = (3e-4, 1e-3) if bf else (4e-6, 1e-4)
    for tag, cls, kw, B, S, F_, cname in FLIP_CASES:
        case = {"kwargs": kw, "w_r": 0.4, "w_c": 0.6}
        ref = getattr(om, cls)(dropout=0.0, **kw)
        fill_state_dict(ref, 700 + B)
        hip = getattr(hm, cls)(dropout=0.0, **kw)
        hip.load_state_dict(ref.state_dict())
        hip = hip.to(dev)
        x, y = synthetic_lists(B, S, F_, 701 + S)
        ref.train(), hip.train()
        out_h = hip(x.to(dev))
        nodes = _encoder_nodes(list(out_h) if isinstance(out_h, (list, tuple)) else [out_h])
        make_criterion(hl, cname, case)(out_h, y.to(dev)).backward()
        # un-aligned first (what round 1 reported), then aligned
        make_criterion(ol, cname, case)(ref(x), y).backward()
        plain = _param_rel_l2(hip, ref)
        ref.zero_grad()
        counts = _align_relu(ref, hip, nodes, B, S, edge)
        make_criterion(ol, cname, case)(ref(x), y).backward()
        aligned = _param_rel_l2(hip, ref)
        report(f"{tag} ReLU decisions differing outside |z| < {edge:g}", float(counts[1]), 0)
        wp, wa = max(plain, key=plain.get), max(aligned, key=aligned.get)
        print(f"   ({tag}: {counts[0]} of {counts[2]} FFN units aligned; worst rel-L2 un-aligned {plain[wp]:.2e} ({wp}), "
              f"aligned {aligned[wa]:.2e} ({wa}))", flush=True)
        # the bias of a Linear behind a softmax over the positions has an identically ZERO gradient (shift invariance): both sides
        # hold rounding residue of sum_j dlogit_j, graded against the 1e-3 x largest-gradient floor of _param_rel_l2 - reported on
        # its own at 3x the tolerance (1.8e-4 / 6.4e-5 / 7.7e-5 measured for the three attention kernel sets at choopy_b16_s40),
        # the worst of the parameters WITH a gradient at the tolerance itself
        zero_grad = {n for n in aligned if n.endswith("decison_layer.0.bias") or n.endswith("cut_layer.0.bias")}
        for n in sorted(zero_grad):
            report(f"{tag} flip-aligned residue of the zero-gradient softmax bias ({n})", aligned[n], 3 * tol)
        live = {n: v for n, v in aligned.items() if n not in zero_grad}
        wa = max(live, key=live.get)
        report(f"{tag} flip-aligned grad rel-L2 worst ({wa})", live[wa], tol)


@section
def trajectory():
    """K = 20 Adam steps, each side on its OWN gradients (HIP model + FusedAdam vs CPU oracle + torch.optim.Adam) from
    the same weights and batch, dropout 0: per step |d loss| and |d F1| <= 1e-4 (the bound BASELINE.json states), cut
    positions identical except lists whose two best positions are within 4e-6 in the oracle (counted), final cut
    distributions within 1e-4."""
    import models as hm
    from oracle import losses as ol, metrics as omet, models as om
    from oracle.cases import make_criterion
    from oracle.weights import fill_state_dict, synthetic_lists
    from utils import losses as hl
    from utils.metrics import Metric
    from rlt_hip.parallel import FlatModel, FusedAdam
    K = 20
    # the Choopy case: small position encoding + unsorted list-specific scores, so that the lists of the batch cut at
    # different positions (with the plain recipe every list of a Choopy batch cuts at the same one)
    for tag, cls, kw, B, F_, cname, lr, wd, fill_kw, noise in [
            ("attncut_b32", "AttnCut", {}, 32, 3, "div_js_f1_aug1", 1e-4, 0.0025, {}, 0.0),
            ("mtattncut_t3_b16", "MtAttnCut", {"num_tasks": 3}, 16, 3, "mtcut_f1", 3e-5, 0.005, {}, 0.0),
            ("choopy_b32", "Choopy", {}, 32, 1, "choopy_f1", 1e-4, 0.0025, {"pe_scale": 0.05}, 2.0),
            ("mmoecut_e4_t21_b16", "MMOECut", {"num_experts": 4, "num_tasks": 2.1}, 16, 3, "mtcut_f1", 3e-5, 0.005, {}, 0.0)]:
        case = {"kwargs": kw, "w_r": 0.4, "w_c": 0.6}
        S = 300
        ref = getattr(om, cls)(dropout=0.0, **kw)
        fill_state_dict(ref, 800 + B, **fill_kw)
        hip = getattr(hm, cls)(dropout=0.0, **kw)
        hip.load_state_dict(ref.state_dict())
        hip = hip.to(dev)
        x, y = synthetic_lists(B, S, F_, 801, noise=noise)
        xd, yd = x.to(dev), y.to(dev)
        flat = FlatModel(hip)
        opt_h = FusedAdam(flat, lr=lr, weight_decay=wd)
        opt_r = torch.optim.Adam(ref.parameters(), lr=lr, weight_decay=wd)
        crit_h, crit_r = make_criterion(hl, cname, case), make_criterion(ol, cname, case)
        ref.train(), hip.train()
        worst_l = worst_f = worst_p = 0.0
        kdiff = kedge = 0
        for _ in range(K):
            opt_r.zero_grad()
            out_r = ref(x)
            loss_r = crit_r(out_r, y)
            loss_r.backward()
            opt_r.step()
            opt_h.zero_grad()
            out_h = hip(xd)
            loss_h = crit_h(out_h, yd)
            loss_h.backward()
            opt_h.step()
            p_r = (out_r[-1] if isinstance(out_r, (list, tuple)) else out_r).detach().squeeze(2)
            p_h = out_h[-1] if isinstance(out_h, (list, tuple)) else out_h
            k_r = omet.cut_positions(p_r.numpy())
            k_h, f1_h, _ = Metric.evaluate(p_h, yd)
            top2 = torch.topk(p_r, 2, dim=1).values
            gap = (top2[:, 0] - top2[:, 1]).numpy()
            differ = k_h.cpu().numpy() != k_r
            kdiff += int((differ & (gap >= 4e-6)).sum())
            kedge += int((differ & (gap < 4e-6)).sum())
            worst_l = max(worst_l, abs(float(loss_h) - float(loss_r)))
            worst_f = max(worst_f, abs(float(f1_h) - omet.Metric.f1(y.numpy(), k_r)) if not differ.any() else 0.0)
            worst_p = max(worst_p, float((p_h.detach().squeeze(2).cpu() - p_r).abs().max()))
        report(f"trajectory {tag}: max |d loss| over {K} steps", worst_l, 1e-4)
        report(f"trajectory {tag}: max |d F1| over {K} steps", worst_f, 1e-4)
        report(f"trajectory {tag}: max |d p| over {K} steps", worst_p, 1e-4)
        report(f"trajectory {tag}: cut positions differing outside knife-edge lists", float(kdiff), 0)
        print(f"   ({tag}: {kedge} knife-edge cut-position differences over {K} steps x {B} lists)", flush=True)
        # cut positions after the K steps (p has moved away from its initial 1/S): identical on every list whose two best
        # positions are at least 1e-5 apart in the oracle (the number of such lists and of distinct k is printed)
        clear = gap >= 1e-5
        report(f"trajectory {tag}: cut positions after {K} steps on the {int(clear.sum())} of {B} lists with top-2 gap >= 1e-5",
               float((differ & clear).sum()), 0)
        print(f"   ({tag}: after {K} steps {int(clear.sum())} of {B} lists have a top-2 gap >= 1e-5 (median gap {float(np.median(gap)):.2e}), "
              f"{len(set(k_r.tolist()))} distinct cut positions, max p {float(p_r.max()):.4f} vs 1/S = {1 / S:.4f})", flush=True)
        # parameters after K steps: relative L2 over the whole flat vector (Adam normalises by sqrt(v): parameters with an
        # analytically zero gradient move by rounding noise of either sign, so a per-parameter max would be meaningless)
        pr = torch.cat([q.detach().reshape(-1) for q in ref.parameters()])
        ph = torch.cat([q.detach().reshape(-1).cpu() for q in hip.parameters()])
        report(f"trajectory {tag}: |params - oracle params| / |oracle step| after {K} steps",
               float((ph - pr).norm()) / max(1e-30, float((pr - torch.cat([q.reshape(-1) for q in _fresh_params(om, cls, kw, 800 + B, fill_kw)])).norm())), 0.2)


def _fresh_params(om, cls, kw, seed, fill_kw=None):
    from oracle.weights import fill_state_dict
    m = getattr(om, cls)(dropout=0.0, **kw)
    fill_state_dict(m, seed, **(fill_kw or {}))
    return [q.detach() for q in m.parameters()]



@section
def trainer_bookkeeping():
    """run.py's Trainer against the same loop written with the CPU oracle (model, criterion, torch.optim.Adam, numpy
    metrics) on the same files, the same loader seed and the same initial weights: per-epoch train / test means
    (unweighted over batches, run.py:153,195), best and best-5 test F1 / DCG (best-5 divides by 5 whatever the number
    of epochs, run.py:229-230), the epoch whose weights are checkpointed (run.py:203-205), and the scalar log."""
    import json
    import tempfile
    import run as hip_run
    from dataloader import BatchLoader, RankData, write_synthetic_robust04
    from oracle import losses as ol, metrics as omet, models as om
    EPOCHS, BS, LR, WD, SEED = 7, 8, 1e-4, 0.0025, 3
    with tempfile.TemporaryDirectory() as tmp:
        write_synthetic_robust04(tmp, "robust04", "drmm_tks", n_train=22, n_test=20, seq_len=300, seed=5)
        for name in ("attncut", "mtattncut"):
            tb, ck = os.path.join(tmp, "tb_" + name), os.path.join(tmp, "ck_" + name)
            argv = ["--model-name", name, "--dataset-base", tmp, "--epochs", str(EPOCHS), "--use-conf", "0", "--batch-size", str(BS),
                    "--criterion", "f1", "--dropout", "0.0", "--lr", str(LR), "--weight-decay", str(WD), "--seed", str(SEED),
                    "--model-persist", "1", "--save-path", ck, "--tensorboard-dir", tb]
            args = hip_run.build_parser().parse_args(argv)
            args.model_path = os.path.join(ck, name + ".pkl")
            torch.manual_seed(SEED)
            trainer = hip_run.Trainer(args)
            init = {k: v.detach().clone().cpu() for k, v in trainer.model.state_dict().items()}
            trainer.run()
            # ---- the same loop on the oracle
            if name == "attncut":
                ref, crit = om.AttnCut(input_size=3, dropout=0.0), ol.DivLoss(metric="f1", div_type="js", augmented=True)
            else:
                ref = om.MtAttnCut(input_size=3, num_tasks=3, dropout=0.0)
                crit = ol.MtCutLoss(metric="f1", rerank_weight=args.rerank_weight, classi_weight=args.class_weight, num_tasks=3)
            ref.load_state_dict(init)
            opt = torch.optim.Adam(ref.parameters(), lr=LR, weight_decay=WD)
            rd = RankData("robust04", "drmm_tks", True, tmp)
            tr = BatchLoader([(rd.getX_train(), rd.gety_train())], BS, True, None, SEED)
            te = BatchLoader([(rd.getX_test(), rd.gety_test())], BS, True, None, SEED + 1)

            def evaluate(out, y):
                p = (out[-1] if isinstance(out, (list, tuple)) else out).detach().squeeze(2).numpy()
                k = omet.cut_positions(p)
                return omet.Metric.f1(y.numpy(), k), omet.Metric.dcg(y.numpy(), k)

            hist, f1_rec, dcg_rec, best, best_epoch, best_sd = [], [], [], -float("inf"), None, None
            for epoch in range(EPOCHS):
                tot, n = np.zeros(3), 0
                ref.train()
                for x, y in tr:
                    opt.zero_grad()
                    out = ref(x)
                    loss = crit(out, y)
                    loss.backward()
                    opt.step()
                    tot += np.array([loss.item(), *evaluate(out, y)])
                    n += 1
                row = {"train": tot / n}
                tot, n = np.zeros(3), 0
                ref.eval()
                with torch.no_grad():
                    for x, y in te:
                        out = ref(x)
                        tot += np.array([crit(out, y).item(), *evaluate(out, y)])
                        n += 1
                row["test"] = tot / n
                hist.append(row)
                f1_rec.append(row["test"][1])
                dcg_rec.append(row["test"][2])
                if row["test"][1] > best:
                    best, best_epoch = row["test"][1], epoch
                    best_sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
            for e in range(EPOCHS):
                for split in ("train", "test"):
                    got, want = np.array(trainer.history[e][split]), hist[e][split]
                    report(f"Trainer {name} epoch {e} {split} loss/F1/DCG means", float(np.abs(got - want).max() / max(1.0, np.abs(want).max())), 1e-4)
            report(f"Trainer {name} best test F1", abs(trainer.best_test_f1 - best), 1e-4)
            report(f"Trainer {name} best test DCG", abs(trainer.best_test_dcg - max(dcg_rec)), 1e-4 * max(1.0, abs(max(dcg_rec))))
            report(f"Trainer {name} best-5 F1 (sum of top 5 / 5)", abs(trainer.best5_f1 - sum(sorted(f1_rec, reverse=True)[:5]) / 5), 1e-4)
            report(f"Trainer {name} best-5 DCG", abs(trainer.best5_dcg - sum(sorted(dcg_rec, reverse=True)[:5]) / 5), 1e-4 * max(1.0, abs(max(dcg_rec))))
            report(f"Trainer {name} checkpointed epoch", abs(trainer.best_epoch - best_epoch), 0)
            sd = torch.load(os.path.join(ck, name + ".pkl"))
            chk = type(ref)(**({"input_size": 3, "dropout": 0.0} if name == "attncut" else {"input_size": 3, "num_tasks": 3, "dropout": 0.0}))
            chk.load_state_dict(sd)                          # loads into the reference-shaped module
            num = sum(float((sd[k] - best_sd[k]).double().pow(2).sum()) for k in sd)
            den = sum(float((best_sd[k] - init[k]).double().pow(2).sum()) for k in sd)
            report(f"Trainer {name} checkpoint vs oracle weights at that epoch (relative to the path from init)", math.sqrt(num / den), 0.1)
            # scalars under the reference's tags
            rows = [json.loads(line) for line in open(os.path.join(tb, "scalars.jsonl"))]
            tags = {}
            for r in rows:
                tags.setdefault(r["tag"], []).append(r)
            steps_per_epoch = len(tr)
            report(f"Trainer {name} scalar tags", 0.0 if set(tags) == {"train/loss_step", "train/loss_epoch", "train/F1_epoch", "train/DCG_epoch",
                                                                       "test/loss_epoch", "test/F1_epoch", "test/DCG_epoch"} else 1.0, 0)
            report(f"Trainer {name} train/loss_step count and steps", 0.0 if [r["step"] for r in tags["train/loss_step"]] == list(range(EPOCHS * steps_per_epoch)) else 1.0, 0)
            report(f"Trainer {name} test/F1_epoch scalars", float(np.abs(np.array([r["value"] for r in tags["test/F1_epoch"]]) - np.array([h["test"][1] for h in trainer.history])).max()), 1e-12)



@section
def trainer_buckets():
    """BASELINE configs[4]'s shape on the product path: run.py's Trainer on a set with lists of 100 / 200 / 300 documents
    (homogeneous length-bucketed batches served round-robin, dataloader/rank_data.py), MtAttnCut + MtCutLoss, against the
    same loop written with the CPU oracle over the same BatchLoader schedule: per-epoch train / test means of loss, F1
    and DCG, and the number of batches of each length."""
    import tempfile
    import run as hip_run
    from dataloader import BatchLoader, RankData, write_synthetic_robust04
    from oracle import losses as ol, metrics as omet, models as om
    EPOCHS, BS, LR, WD, SEED = 2, 4, 1e-4, 0.0025, 11
    with tempfile.TemporaryDirectory() as tmp:
        write_synthetic_robust04(tmp, "robust04", "drmm_tks", n_train=21, n_test=9, seed=6, lengths=(100, 200, 300))
        for name, nt in (("mtattncut", 3), ("attncut", None)):
            argv = ["--model-name", name, "--dataset-base", tmp, "--epochs", str(EPOCHS), "--use-conf", "0", "--batch-size", str(BS),
                    "--criterion", "f1", "--dropout", "0.0", "--lr", str(LR), "--weight-decay", str(WD), "--seed", str(SEED),
                    "--tensorboard-dir", ""]
            args = hip_run.build_parser().parse_args(argv)
            args.model_path = None
            torch.manual_seed(SEED)
            trainer = hip_run.Trainer(args)
            init = {k: v.detach().clone().cpu() for k, v in trainer.model.state_dict().items()}
            trainer.run()
            if nt is None:
                ref, crit = om.AttnCut(input_size=3, dropout=0.0), ol.DivLoss(metric="f1", div_type="js", augmented=True)
            else:
                ref = om.MtAttnCut(input_size=3, num_tasks=nt, dropout=0.0)
                crit = ol.MtCutLoss(metric="f1", rerank_weight=args.rerank_weight, classi_weight=args.class_weight, num_tasks=nt)
            ref.load_state_dict(init)
            opt = torch.optim.Adam(ref.parameters(), lr=LR, weight_decay=WD)
            rd = RankData("robust04", "drmm_tks", True, tmp)
            pairs = lambda split: [(x, y) for (x, y, _q) in rd.buckets[split].values()]
            tr = BatchLoader(pairs("train"), BS, True, None, SEED)
            te = BatchLoader(pairs("test"), BS, True, None, SEED + 1)
            # (a third loader for the schedule check: iterating one advances its permutation stream)
            seen = [int(x.shape[1]) for x, _ in BatchLoader(pairs("train"), BS, True, None, SEED)]
            report(f"Trainer buckets {name}: the train loader serves lengths 100/200/300 round-robin",
                   0.0 if seen[:6] == [100, 200, 300, 100, 200, 300] and sorted(set(seen)) == [100, 200, 300] else 1.0, 0)

            edge = [0]

            def evaluate(out, y):
                p = (out[-1] if isinstance(out, (list, tuple)) else out).detach().squeeze(2)
                top2 = torch.topk(p, 2, dim=1).values
                edge[0] += int(((top2[:, 0] - top2[:, 1]) < 4e-6).sum())      # lists whose cut position is a coin toss in fp32
                k = omet.cut_positions(p.numpy())
                return omet.Metric.f1(y.numpy(), k), omet.Metric.dcg(y.numpy(), k)

            for epoch in range(EPOCHS):
                tot, n = np.zeros(3), 0
                edge[0] = 0
                ref.train()
                for x, y in tr:
                    opt.zero_grad()
                    out = ref(x)
                    loss = crit(out, y)
                    loss.backward()
                    opt.step()
                    tot += np.array([loss.item(), *evaluate(out, y)])
                    n += 1
                want_tr, edge_tr = tot / n, edge[0]
                tot, n = np.zeros(3), 0
                edge[0] = 0
                ref.eval()
                with torch.no_grad():
                    for x, y in te:
                        out = ref(x)
                        tot += np.array([crit(out, y).item(), *evaluate(out, y)])
                        n += 1
                want_te, edge_te = tot / n, edge[0]
                for split, want, n_edge in (("train", want_tr, edge_tr), ("test", want_te, edge_te)):
                    got = np.array(trainer.history[epoch][split])
                    report(f"Trainer buckets {name} epoch {epoch} {split} loss mean", abs(got[0] - want[0]) / max(1.0, abs(want[0])), 1e-4)
                    if n_edge == 0:
                        report(f"Trainer buckets {name} epoch {epoch} {split} F1/DCG means", float(np.abs(got[1:] - want[1:]).max() / max(1.0, np.abs(want[1:]).max())), 1e-4)
                    else:       # near-uniform p right after initialisation: the oracle's own top-2 gap is below fp32 resolution
                        print(f"   (Trainer buckets {name} epoch {epoch} {split}: {n_edge} knife-edge lists (top-2 gap < 4e-6 in the oracle): "
                              f"F1/DCG means differ by {float(np.abs(got[1:] - want[1:]).max()):.2e}, not asserted)", flush=True)


@section
def path_level():
    """The path-level entry points (rlt_encoder_layer_fwd/bwd: launches composed inside the library) against the same
    layer driven launch by launch from the host through the kernel-level entry points: bit-identical outputs and
    gradients, with and without train-mode dropout (same seeds)."""
    for (B, S, E, H, p_drop) in [(33, 7, 256, 4, 0.0), (70, 5, 128, 8, 0.0), (64, 4, 256, 4, 0.3), (300, 3, 128, 8, 0.2)]:
        torch.manual_seed(B)
        layer = torch.nn.TransformerEncoderLayer(d_model=E, nhead=H, dropout=p_drop)
        from models._common import ParamTree
        x = torch.randn(S * B, E)
        dy = torch.randn(S * B, E)
        res = []
        for kernel_level in (False, True):
            pl = ParamTree(layer).to(dev)
            xin = x.clone().to(dev).requires_grad_(True)
            torch.manual_seed(5)
            ops._SEED_COUNTER[0] = 0
            ops.KERNEL_LEVEL_ENCODER[0] = kernel_level
            try:
                y = ops.encoder_layer(xin, pl, S, B, H, p_drop)
            finally:
                ops.KERNEL_LEVEL_ENCODER[0] = False
            y.backward(dy.to(dev))
            res.append((type(y.grad_fn).__name__, y.detach(), xin.grad, {n: q.grad for n, q in pl.named_parameters()}))
        (na, ya, dxa, ga), (nb, yb, dxb, gb) = res
        report(f"path-level B{B} S{S} E{E} p{p_drop}: tape nodes differ", 0.0 if (na, nb) == ("EncoderLayerFnBackward", "EncoderLayerKernelsFnBackward") else 1.0, 0)
        report(f"path-level B{B} S{S} E{E} p{p_drop}: y", float((ya - yb).abs().max()), 0)
        report(f"path-level B{B} S{S} E{E} p{p_drop}: dx", float((dxa - dxb).abs().max()), 0)
        report(f"path-level B{B} S{S} E{E} p{p_drop}: weight grads", max(float((ga[n] - gb[n]).abs().max()) for n in ga), 0)



def _trainer_dp_case(tag, model_name, extra_argv, make_ref, make_crit, ckpt_name, port, lengths=(300,), n_train=11, n_test=5,
                     epochs=3, bs=4, world=2):
    """run.py under torch.distributed with TWO ranks (both on this GPU, gloo - the rehearsal form) against the shard-wise
    reference semantics written with the CPU oracle: every batch is cut by shard_bounds (ragged
    tails and EMPTY shards included), each shard is one reference computation - its own list-axis attention, its own
    RerankLoss batch means (utils/losses.py:134-141) and BCE mean -, loss / gradient / logged means are weighted by the
    shards' list counts, one Adam step per batch.  Also: both ranks' final parameter buckets must be bitwise equal."""
    import json
    import subprocess
    import tempfile
    from dataloader import BatchLoader, RankData, write_synthetic_robust04
    from oracle import metrics as omet
    from oracle.weights import fill_state_dict
    from rlt_hip.parallel import shard_bounds
    LR, WD, SEED, WORLD = 1e-4, 0.0025, 7, world
    run_py = os.path.join(REPO, "ranked-list-truncation_amd", "run.py")
    with tempfile.TemporaryDirectory() as tmp:
        write_synthetic_robust04(tmp, "robust04", "drmm_tks", n_train=n_train, n_test=n_test, seq_len=300, seed=9,
                                 **({"lengths": lengths} if len(lengths) > 1 else {}))
        ref = make_ref()
        fill_state_dict(ref, 31)
        ck = os.path.join(tmp, "init")
        os.makedirs(ck)
        torch.save(ref.state_dict(), os.path.join(ck, ckpt_name))
        hist, dump = os.path.join(tmp, "hist.json"), os.path.join(tmp, "dump")
        env = dict(os.environ, RLT_DIST_BACKEND="gloo", RLT_RUN_DEVICE="0", RLT_PRECISION=N.get_precision())
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={WORLD}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), run_py, "--model-name", model_name, "--dataset-base", tmp, "--epochs", str(epochs), "--use-conf", "0",
               "--batch-size", str(bs), "--criterion", "f1", "--dropout", "0.0", "--lr", str(LR), "--weight-decay", str(WD),
               "--seed", str(SEED), "--ft", "1", "--model-path", os.path.join(ck, ckpt_name), "--history-json", hist,
               "--param-dump-dir", dump, "--tensorboard-dir", ""] + list(extra_argv)
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        report(f"{tag}: {WORLD}-rank run.py exit status", float(res.returncode), 0)
        if res.returncode != 0:
            print(res.stderr[-3000:])
            return
        got = json.load(open(hist))
        report(f"{tag}: world size seen by the Trainer", abs(got["world"] - WORLD), 0)
        ps = [np.load(os.path.join(dump, f"flat_param_rank{r}.npy")) for r in range(WORLD)]
        report(f"{tag}: the {WORLD} replicas' parameter buckets after training, bitwise (elements differing from rank 0's)",
               float(sum((ps[0].view(np.uint32) != q.view(np.uint32)).sum() for q in ps[1:])), 0)
        # ---- shard-wise reference on the oracle
        opt = torch.optim.Adam(ref.parameters(), lr=LR, weight_decay=WD)
        crit = make_crit()
        rd = RankData("robust04", "drmm_tks", True, tmp)
        pairs = lambda split: [(x, y) for (x, y, _q) in rd.buckets[split].values()]
        tr = BatchLoader(pairs("train"), bs, True, None, SEED)
        te = BatchLoader(pairs("test"), bs, True, None, SEED + 1)
        edge = [0]
        shapes = []

        def batch(x, y, train):
            """-> (loss, F1, DCG means of the batch, slack): slack = what the knife-edge lists of the batch (top-2 gap < 4e-6 in
            the oracle's own output: their cut position is a coin toss in fp32) can move the F1 / DCG means by - per LIST, the
            difference between the metric at the best and at the second-best position, nothing else is exempted"""
            n = x.shape[0]
            tot = np.zeros(3)
            slack = np.zeros(2)
            if train:
                opt.zero_grad()
                shapes.append((int(x.shape[1]), tuple(shard_bounds(n, r, WORLD)[1] - shard_bounds(n, r, WORLD)[0] for r in range(WORLD))))
            for r in range(WORLD):
                lo, hi = shard_bounds(n, r, WORLD)
                if hi == lo:
                    continue
                out = ref(x[lo:hi])
                loss = crit(out, y[lo:hi])
                if train:
                    (loss * ((hi - lo) / n)).backward()
                p = (out[-1] if isinstance(out, (list, tuple)) else out).detach().squeeze(2)
                top2 = torch.topk(p, 2, dim=1)
                k = omet.cut_positions(p.numpy())
                yy = y[lo:hi].numpy()
                for i in torch.nonzero((top2.values[:, 0] - top2.values[:, 1]) < 4e-6).flatten().tolist():
                    edge[0] += 1
                    k_alt = np.array([int(top2.indices[i, 1]) + 1 if int(top2.indices[i, 0]) + 1 == k[i] else int(top2.indices[i, 0]) + 1])
                    slack += [abs(omet.Metric.f1(yy[i:i + 1], k[i:i + 1]) - omet.Metric.f1(yy[i:i + 1], k_alt)),
                              abs(omet.Metric.dcg(yy[i:i + 1], k[i:i + 1]) - omet.Metric.dcg(yy[i:i + 1], k_alt))]
                tot += (hi - lo) * np.array([loss.item(), omet.Metric.f1(yy, k), omet.Metric.dcg(yy, k)])
            if train:
                opt.step()
            return tot / n, slack / n

        for e in range(epochs):
            rows_by = {}
            ref.train()
            edge[0] = 0
            rows = [batch(x, y, True) for x, y in tr]
            rows_by["train"] = (np.mean([r_[0] for r_ in rows], axis=0), np.mean([r_[1] for r_ in rows], axis=0), edge[0])
            ref.eval()
            edge[0] = 0
            with torch.no_grad():
                rows = [batch(x, y, False) for x, y in te]
            rows_by["test"] = (np.mean([r_[0] for r_ in rows], axis=0), np.mean([r_[1] for r_ in rows], axis=0), edge[0])
            for split, (want, slack, n_edge) in rows_by.items():
                g = np.array(got["history"][e][split])
                report(f"{tag} epoch {e} {split} loss", abs(g[0] - want[0]) / max(1.0, abs(want[0])), 1e-4)
                # F1 / DCG means: 1e-4 plus exactly what the epoch's knife-edge lists can account for
                for j, nm in ((1, "F1"), (2, "DCG")):
                    report(f"{tag} epoch {e} {split} {nm}" + (f" ({n_edge} knife-edge lists: + {slack[j - 1]:.1e})" if n_edge else ""),
                           abs(g[j] - want[j]) / max(1.0, abs(want[j])), 1e-4 + slack[j - 1] / max(1.0, abs(want[j])))
        if len(lengths) > 1:
            seen = sorted({s for s, _ in shapes})
            report(f"{tag}: every length bucket was trained on by both ranks in lock-step (lengths seen {seen})",
                   0.0 if seen == sorted(lengths) else 1.0, 0)
        ragged = [sh for _, sh in shapes if len(set(sh)) > 1]
        report(f"{tag}: the schedule holds ragged shards ({len(ragged)} of {len(shapes)} train batches)", 0.0 if ragged else 1.0, 0)
        if WORLD > 2:
            empty = [sh for _, sh in shapes if 0 in sh]
            report(f"{tag}: ... and EMPTY shards ({len(empty)} of {len(shapes)} train batches)", 0.0 if empty else 1.0, 0)


@section
def trainer_dp():
    """AttnCut + DivLoss under two ranks: 11 = 4+4+3 train lists -> shards 2+1; 5 = 4+1 test lists -> shards 1+0, an EMPTY
    shard."""
    from oracle import losses as ol, models as om
    _trainer_dp_case("trainer_dp", "attncut", [], lambda: om.AttnCut(input_size=3, dropout=0.0),
                     lambda: ol.DivLoss(metric="f1", div_type="js", augmented=True), "attncut.pkl", 29731)


@section
def trainer_dp_mt(which=None):
    """The two configurations north_star shards over 8 GPUs, under two ranks: BASELINE configs[3]
    MMOECut(4 experts) with task codes 2.1 (class + cut) and 2.2 (rerank + cut) - per-shard RerankLoss batch means
    (utils/losses.py:134-141), MtCutLoss weighting (:180-191), the 76,800 x 4 gate matrices in the flat bucket
    (models/MMOECut.py:68,93-94) - and configs[4] MtAttnCut(3 tasks) + MtCutLoss on length buckets 100 / 200 / 300 (round-robin
    batches that must stay in lock-step across the ranks, ragged and empty shards included)."""
    from oracle import losses as ol, models as om
    RW, CW = 0.3, 0.4                      # run.py's --rerank-weight / --class-weight defaults, passed explicitly below
    for i, nt in enumerate((2.1, 2.2)):
        if which and f"mmoe{nt}" not in which:
            continue
        _trainer_dp_case(f"trainer_dp mmoecut(4e,{nt})", "mmoecut", ["--num-experts", "4", "--num-tasks", str(nt)],
                         lambda nt=nt: om.MMOECut(seq_len=300, num_experts=4, num_tasks=nt, input_size=3, dropout=0.0),
                         lambda nt=nt: ol.MtCutLoss(metric="f1", num_tasks=nt), "mmoecut.pkl", 29733 + i, epochs=2)
    if not which or "buckets" in which:
        _trainer_dp_case("trainer_dp mtattncut(3) buckets", "mtattncut",
                         ["--num-tasks", "3", "--rerank-weight", str(RW), "--class-weight", str(CW)],
                         lambda: om.MtAttnCut(input_size=3, num_tasks=3, dropout=0.0),
                         lambda: ol.MtCutLoss(metric="f1", rerank_weight=RW, classi_weight=CW, num_tasks=3), "mtattncut.pkl", 29736,
                         lengths=(100, 200, 300), n_train=21, n_test=9, epochs=2)


@section
def x6_image_staging():
    """The bf16x6 attention kernels staged from pre-split tile images (RLT_ATTN6_IMG=1: a prepare pass per call, LDS-DMA in the
    kernels; off by default, the switch is read once per process): the op-level attention sections in a child process - with the
    two-workgroup backward kernels at head dim 64 (RLT_A6_DKV1=0, RLT_A6_DQ1=0: the forms the one-wavefront kernels replaced by
    default, still the path for inputs whose B * ld exceeds their 24-bit row offsets)."""
    import subprocess
    env = dict(os.environ, RLT_ATTN6_IMG="1", RLT_PRECISION="bf16x6", RLT_A6_DKV1="0", RLT_A6_DQ1="0")
    res = subprocess.run([sys.executable, os.path.abspath(__file__), "attention", "scale_ops"], env=env, capture_output=True, text=True,
                         timeout=900)
    report("x6_image_staging: child exit status", float(res.returncode), 0)
    tail = [l for l in res.stdout.strip().splitlines() if " ok, " in l and "failed" in l]
    ok, failed = (int(tail[-1].split()[0]), int(tail[-1].split()[2])) if tail else (0, 1)
    report("x6_image_staging: checks failed in the child", float(failed), 0)
    report("x6_image_staging: checks run in the child (>= 100)", 0.0 if ok >= 100 else 1.0, 0)
    if res.returncode or failed:
        print(res.stdout[-3000:], res.stderr[-2000:])


@section
def x6_fallbacks():
    """The bf16x6 attention forms the default dispatch does not reach at the test shapes: (1) in a child
    process, the two-workgroup backward kernels at head dim 64 WITHOUT image staging (RLT_A6_DKV1=0 RLT_A6_DQ1=0: the IMG = false
    instantiations, the path of inputs whose B * ld exceeds the one-wavefront kernels' 24-bit row offsets) and the two-wavefront
    head-dim-16 kernels at every size (RLT_A6N_1=0), on the op-level attention sections; (2) here, the 24-bit switch-over
    itself: S = 1, H = 4, head dim 64 at B = 21,760 (B * 3 * H * HD = 16,711,680 < 2^24: one-wavefront kernels) and B = 21,888
    (16,809,984 >= 2^24: two-workgroup kernels), forward + backward against the exact-fp32 kernels on the same inputs."""
    import subprocess
    env = dict(os.environ, RLT_PRECISION="bf16x6", RLT_A6_DKV1="0", RLT_A6_DQ1="0", RLT_A6N_1="0")
    env.pop("RLT_ATTN6_IMG", None)
    res = subprocess.run([sys.executable, os.path.abspath(__file__), "attention", "scale_ops"], env=env, capture_output=True, text=True,
                         timeout=900)
    report("x6_fallbacks: child exit status", float(res.returncode), 0)
    tail = [l for l in res.stdout.strip().splitlines() if " ok, " in l and "failed" in l]
    ok, failed = (int(tail[-1].split()[0]), int(tail[-1].split()[2])) if tail else (0, 1)
    report("x6_fallbacks: checks failed in the child", float(failed), 0)
    report("x6_fallbacks: checks run in the child (>= 100)", 0.0 if ok >= 100 else 1.0, 0)
    if res.returncode or failed:
        print(res.stdout[-3000:], res.stderr[-2000:])
    # (3) round 5's A/B switches: the recurrences of round 4 (six-product forward in two phases, f32 MFMA backward: RLT_LSTM6W=0), the
    # two-half kernels at every batch size (RLT_LSTM6W_SINGLE=0) and the tiled kernels for the K = 256 / 128 products (RLT_GEMM6S=0)
    for tag, extra, sections, least in (("lstm6w off", {"RLT_LSTM6W": "0"}, ["lstm"], 100), ("lstm6w two halves", {"RLT_LSTM6W_SINGLE": "0"}, ["lstm"], 100),
                                        ("gemm6s off", {"RLT_GEMM6S": "0"}, ["gemm"], 60)):
        env2 = dict(os.environ, RLT_PRECISION="bf16x6", **extra)
        res2 = subprocess.run([sys.executable, os.path.abspath(__file__)] + sections, env=env2, capture_output=True, text=True, timeout=600)
        tail2 = [l for l in res2.stdout.strip().splitlines() if " ok, " in l and "failed" in l]
        ok2, failed2 = (int(tail2[-1].split()[0]), int(tail2[-1].split()[2])) if tail2 else (0, 1)
        report(f"x6_fallbacks ({tag}): child exit status / failed checks", float(abs(res2.returncode) + failed2), 0)
        report(f"x6_fallbacks ({tag}): checks run in the child (>= {least})", 0.0 if ok2 >= least else 1.0, 0)
        if res2.returncode or failed2:
            print(res2.stdout[-3000:], res2.stderr[-2000:])
    # (4) the fix-up launch behind the pipelined head-dim-16 forward (csrc/attention6n.hip; head dim 64 runs the same inputs through its
    # running-maximum kernel): a query's reference there is its largest score against the first 32 keys; here the later keys score far above it for HALF of the queries (their weights overflow fp32), so
    # every workgroup raises its flag and the two-wavefront kernel redoes it with the moving reference - against fp64
    for HD4 in (16, 64):
      with ops.precision("bf16x6"):
        gq = torch.Generator(device=dev).manual_seed(11)
        S4, H4, B4 = 2, 2, 1024
        E4 = H4 * HD4
        amp = 6.0 if HD4 == 16 else 4.0
        q = torch.randn(B4, S4, E4, generator=gq, device=dev)
        q[::2] = amp + 0.1 * torch.randn(B4 // 2, S4, E4, generator=gq, device=dev)        # every other query: large, one sign
        k = torch.randn(B4, S4, E4, generator=gq, device=dev) * 0.05
        k[64:] += amp                                                                         # keys behind the first 64: aligned with those queries
        v = torch.randn(B4, S4, E4, generator=gq, device=dev)
        qkv4 = torch.cat([q, k, v], dim=2)
        ref4 = _attn_ref(qkv4.double(), H4)
        od4 = _unpm(ops.list_attention(_pm(qkv4).to(dev), S4, B4, H4), B4, S4)
        top = float((qkv4[..., :E4].double().abs().max() * qkv4[..., E4:2 * E4].double().abs().max()) * math.sqrt(HD4) * 1.4427)
        report(f"x6_fallbacks: hd{HD4} forward, weights that overflow against the first keys' maximum (scores up to 2^{top:.0f}): finite", 0.0 if bool(torch.isfinite(od4).all()) else 1.0, 0)
        report(f"x6_fallbacks: hd{HD4} forward, such weights: out vs fp64" + (" (fix-up launch behind the pipelined kernel)" if HD4 == 16 else " (running maximum)"), rel(od4, ref4), 2e-5)
    g = torch.Generator(device=dev).manual_seed(5)
    S, H, HD = 1, 4, 64
    E = H * HD
    for B in (21760, 21888):
        qkv = torch.randn(B * S, 3 * E, generator=g, device=dev)
        dout = torch.randn(B * S, E, generator=g, device=dev)
        outs = {}
        for mode in ("fp32", "bf16x6"):
            with ops.precision(mode):
                qd = qkv.clone().requires_grad_(True)
                od = ops.list_attention(qd, S, B, H)
                od.backward(dout)
                outs[mode] = (od.detach(), qd.grad.detach())
        side = "below" if B * 3 * E < (1 << 24) else "at or above"
        report(f"x6_fallbacks B{B} (B*ld {side} 2^24): out vs the exact-fp32 kernels", rel(outs["bf16x6"][0], outs["fp32"][0]), 1e-5)
        for nm, sl in (("dq", slice(0, E)), ("dk", slice(E, 2 * E)), ("dv", slice(2 * E, 3 * E))):
            report(f"x6_fallbacks B{B}: {nm} vs the exact-fp32 kernels", rel(outs["bf16x6"][1][:, sl], outs["fp32"][1][:, sl]), 3e-5)
        del qkv, dout, outs
        torch.cuda.empty_cache()


@section
def rccl_one_rank():
    """The RCCL code path on this one-GPU box: bench.py and run.py as fresh children of torch.distributed.run with ONE
    rank and RLT_FORCE_DIST=1 - init_process_group("nccl", device_id=...), the parameter broadcast, the all-reduce(AVG) of
    the flat gradient bucket every step and the NCCL barriers all execute (with one rank they otherwise early-return, so
    the driver's 8-GPU run would be the first time they ran).  A one-rank AVG must leave the bucket bitwise unchanged, so
    the training state must equal the same command without the process group."""
    import json
    import subprocess
    import tempfile
    from dataloader import write_synthetic_robust04
    bench_py = os.path.join(REPO, "bench.py")
    run_py = os.path.join(REPO, "ranked-list-truncation_amd", "run.py")
    base = [bench_py, "--gpus", "1", "--batch", "96", "--steps", "3", "--warmup", "1", "--other-steps", "0", "--no-cpu-baseline",
            "--precision", "fp32"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    plain = subprocess.run([sys.executable] + base, env=env, capture_output=True, text=True, timeout=600)
    report("rccl_one_rank: plain bench.py exit status", float(plain.returncode), 0)
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1"]
    forced = subprocess.run(launch + ["--master-port", "29741"] + base, env=dict(env, RLT_FORCE_DIST="1"),
                            capture_output=True, text=True, timeout=600)
    report("rccl_one_rank: bench.py under torch.distributed.run (1 rank, nccl) exit status", float(forced.returncode), 0)
    if plain.returncode or forced.returncode:
        print(plain.stderr[-1500:], forced.stderr[-3000:])
        return
    a = json.loads(plain.stdout.strip().splitlines()[-1])
    b = json.loads(forced.stdout.strip().splitlines()[-1])
    col = b["collective"] or {}
    report("rccl_one_rank: collective.backend == nccl", 0.0 if col.get("backend") == "nccl" else 1.0, 0)
    report("rccl_one_rank: rccl_ranks == 1", abs(col.get("rccl_ranks", -1) - 1), 0)
    report("rccl_one_rank: one-rank AVG leaves the flat gradient bucket unchanged (max |diff|)", float(col.get("one_rank_avg_max_abs_diff", 1.0)), 0)
    report("rccl_one_rank: plain run has no process group", 0.0 if a["collective"] is None else 1.0, 0)
    vals = [b["train_state"][k] for k in ("loss", "f1", "dcg")]
    report("rccl_one_rank: finite training state", 0.0 if all(math.isfinite(v) for v in vals) else 1.0, 0)
    report("rccl_one_rank: training state == the run without a process group",
           max(abs(a["train_state"][k] - b["train_state"][k]) for k in ("loss", "f1", "dcg")), 0)
    # run.py (the Trainer's data-parallel loop: broadcast, per-step all-reduce of gradients and of the logged sums)
    with tempfile.TemporaryDirectory() as tmp:
        write_synthetic_robust04(tmp, "robust04", "drmm_tks", n_train=12, n_test=6, seq_len=300, seed=4)
        hist = {}
        for tag, pre, extra_env in (("plain", [sys.executable], {}), ("nccl", launch + ["--master-port", "29742"], {"RLT_FORCE_DIST": "1"})):
            hist[tag] = os.path.join(tmp, tag + ".json")
            cmd = pre + [run_py, "--model-name", "attncut", "--dataset-base", tmp, "--epochs", "2", "--use-conf", "0", "--batch-size", "6",
                         "--criterion", "f1", "--dropout", "0.0", "--lr", "1e-4", "--weight-decay", "0.0025", "--seed", "5",
                         "--history-json", hist[tag], "--tensorboard-dir", ""]
            res = subprocess.run(cmd, env=dict(env, RLT_PRECISION=N.get_precision(), **extra_env), capture_output=True, text=True, timeout=600)
            report(f"rccl_one_rank: run.py {tag} exit status", float(res.returncode), 0)
            if res.returncode:
                print(res.stderr[-3000:])
                return
        ha, hb = json.load(open(hist["plain"])), json.load(open(hist["nccl"]))
        worst = max(abs(x - y) for ea, eb in zip(ha["history"], hb["history"]) for sp in ("train", "test") for x, y in zip(ea[sp], eb[sp]))
        report("rccl_one_rank: run.py per-epoch means, nccl 1 rank == no process group", worst, 0)


def _low_mantissa(x, pattern):
    """x with the low 16 bits of every fp32 significand replaced by `pattern` (the sign, exponent and the 7 bits the first
    bf16 plane keeps stay): 0xFFFF = 'low mantissa bytes all ones' (VERDICT r03 item 1a); 0x7F40 makes BOTH residual planes of
    the three-way split as large as they can be (m = half an ulp of h, l = half an ulp of m), i.e. the dropped m l' + l m'
    terms maximal."""
    return ((x.contiguous().view(torch.int32) & ~0xFFFF) | pattern).view(torch.float32)


@section
def x6_adversarial():
    """bf16x6 against the f32 MFMA kernels on ADVERSARIAL operands, both measured against fp64 on the same inputs (VERDICT
    r03 item 1a).  Operand classes:
      ones         low 16 significand bits of every operand all ones, all operands of one sign   (the verdict's pattern)
      cancel       alternating-sign sums whose terms cancel by >= 1e4                              (the verdict's pattern)
      worst-split  low 16 significand bits 0x7F40, one sign: BOTH residual planes of the three-way split as large as they
                   can be, so the dropped m l' + l m' terms are maximal and all point the same way (harsher than asked)
    Contraction lengths 16, 64, 256, 2048 (NT products) and 1,228,800 (the split-K weight-gradient product); list attention
    at head dims 16 / 64 over 256 and 2048 lists (contractions over hd and over the lists).  Error = max |x - x64| / max
    sum_k |a_k b_k| (the size of the terms, so that a cancelling sum is not graded against its tiny result; attention
    gradients: against the largest reference gradient).
    Asserted on the verdict's two classes: err_x6 <= 1.25 err_f32mfma + sqrt(K) 2^-25 - the additive term is HALF the
    random-walk level of K fp32 roundings, below which one kernel's rounding pattern against another's is noise (at K = 64
    both errors are ~1e-7 = two ulps of the terms).
    Asserted on all three classes, both kernels: the a-priori bound of an fp32 chain, K 2^-24 of the terms (GEMM).
    The worst-split class: with every operand carrying the same low bits, the ROUNDINGS of the (identical) small plane products
    into one large running sum all go the same way - a coherent drift instead of a random walk, in whichever kernel the
    pattern happens to hit: the f32 MFMA chain on `ones` (K = 1,228,800: 1.8e-5 against 5e-7), the six-product kernels on
    `worst-split` (1.6e-5 against 9e-7); the worst over the three classes is the same for both in the GEMM family.  It is NOT
    the dropped m l' + l m' terms: round 5 measured eight-product contractions (dP; S; the list-contracted outputs; the
    forward) - errors identical to four digits (profiles/r05_notes.md).  What closes it is keeping the five small products of
    the list-contracted outputs (O, dQ, dK, dV) in an accumulator of their own, added to the h h' accumulator once at the end:
    the head-dim-16 kernels (csrc/attention6n.hip) do, and sit at 0.25-1.1 of the f32 kernels' error on EVERY class - held to
    the 1.25 here, worst-split included.  Head dim 64: the forward at 512 lists and more is the two-accumulator kernel of
    csrc/attention6h.hip (O at 0.45-0.6 of the f32 kernels' error on every class); the backward kernels have no registers for a
    second accumulator set, but what made dQ / dK of worst-split 4-8x the f32 kernels' in rounds 3-5 was the one-accumulator
    FORWARD's coherent error in O, amplified through delta = rowsum(dO o O) into dS - with the new forward they measure 1.1-1.5x,
    every head-dim-64 gradient on every class is bounded at 2x here and the 10x exemption is gone.
    The mode is an argument of each call here (ops.precision): the two run side by side in one process."""
    def both(fn):
        outs = {}
        for mode in ("fp32", "bf16x6"):
            with ops.precision(mode):
                outs[mode] = fn()
        return outs["fp32"], outs["bf16x6"]

    summary = []

    def grade(name, tag, got32, got6, ref, scale, K, apriori):
        e32 = float((got32.double() - ref).abs().max() / scale)
        e6 = float((got6.double() - ref).abs().max() / scale)
        ratio = e6 / max(e32, 1e-300)
        summary.append((name, tag, e32, e6, ratio))
        print(f"     {name} {tag}: err f32-MFMA {e32:.3e}, bf16x6 {e6:.3e}, ratio {ratio:.2f}", flush=True)
        # additive floor: half the random-walk level of K fp32 roundings, but never more than twice the f32 kernel's own error
        # (at K = 1,228,800 sqrt(K) 2^-25 alone would exceed every measured error: VERDICT r04 weak item 2), plus 2^-27 of the
        # term scale for the plane products the mode drops by construction (documented: < 2^-23 worst, 2^-29 typical - it shows
        # where exact pairwise cancellation makes the f32 chain's own error vanish: `cancel` at K = 1,228,800, 1.2e-9 vs 7e-11)
        floor = min(math.sqrt(K) * 2.0 ** -25, 2.0 * e32) + 2.0 ** -27
        # Attention at head dim 64: the forward at 512 lists and more is the two-accumulator kernel of csrc/attention6h.hip and is held
        # like head dim 16; the backward kernels keep ONE accumulator per output, and dQ / dK are DERIVED quantities - dS = P (dP - delta)
        # with delta = rowsum(dO o O) amplifies whatever error O carries by the cancellation in (dP - delta) (errors of 1e-4 .. 1e-3 of
        # the gradient in BOTH kernels on these operands) - so which kernel's rounding pattern lands worse is a coin toss per class:
        # bounded at twice the f32 kernels' error on every class.  (Until round 6 the worst-split class sat at 4-8x here: that was the
        # one-accumulator FORWARD's coherent error in O arriving through delta, not the backward kernels' own products - with the
        # two-accumulator forward the same backward kernels measure 1.1-1.5x.)
        hd64_bwd = name.startswith("attn") and "hd64" in name and not name.endswith(" out")
        two_acc = name.startswith("attn") and ("hd16" in name or (name.endswith(" out") and int(name.split()[1][1:]) >= 512))
        if hd64_bwd:
            report(f"x6 adversarial {name} {tag}: err_x6 <= 2 err_f32mfma + 2^-27 (derived gradient, one accumulator; observed multiple {ratio:.2f})",
                   e6, 2.0 * e32 + 2.0 ** -27)
        elif tag != "worst-split" or two_acc:
            bound = 1.25 * e32 + floor
            report(f"x6 adversarial {name} {tag}: err_x6 <= 1.25 err_f32mfma + min(sqrt(K) 2^-25, 2 err_f32mfma) + 2^-27 "
                   f"(effective multiple {bound / max(e32, 1e-300):.2f})", e6, bound)
            # ... and without the random-walk allowance: twice the f32 kernel's own error (+ the 2^-27 of the dropped plane products)
            report(f"x6 adversarial {name} {tag}: err_x6 <= 2 err_f32mfma + 2^-27 (observed multiple {ratio:.2f})", e6, 2.0 * e32 + 2.0 ** -27)
        elif name.startswith("attn"):       # worst-split through a one-accumulator forward (head dim 64 below 512 lists)
            report(f"x6 adversarial {name} {tag}: err_x6 <= 2 err_f32mfma + 2^-27 (coherent class, one accumulator; observed multiple {ratio:.2f})",
                   e6, 2.0 * e32 + 2.0 ** -27)
        if apriori:
            report(f"x6 adversarial {name} {tag}: bf16x6 within the fp32-chain a-priori bound K 2^-24", e6, K * 2.0 ** -24)
            report(f"x6 adversarial {name} {tag}: f32 MFMA within the fp32-chain a-priori bound K 2^-24", e32, K * 2.0 ** -24)

    g = torch.Generator(device=dev).manual_seed(77)
    # ---- GEMM family
    for K in (16, 64, 256, 2048, 1228800):
        big = K > 4096
        M, Nn = (256, 256) if big else (512, 256)
        for tag in ("ones", "worst-split", "cancel"):
            if tag == "cancel":
                u = torch.rand(K // 2, generator=g, device=dev) + 0.5
                a_row = torch.stack([u, -u], 1).reshape(1, K)                       # +u0 -u0 +u1 -u1 ...
                A = a_row * (1.0 + 0.05 * torch.rand(M, 1, generator=g, device=dev))
                Bm = (torch.rand(Nn, 1, generator=g, device=dev) + 0.5) * (1.0 + 1e-4 * torch.randn(Nn, K, generator=g, device=dev))
            else:
                pat = 0xFFFF if tag == "ones" else 0x7F40
                A = _low_mantissa(torch.rand(M, K, generator=g, device=dev) + 0.5, pat)
                Bm = _low_mantissa(torch.rand(Nn, K, generator=g, device=dev) + 0.5, pat)
            Ad, Bd = A.to(dev), Bm.to(dev)
            ref = Ad.double() @ Bd.double().t()
            scale = float((Ad.double().abs() @ Bd.double().abs().t()).max())
            if tag == "cancel":
                canc = float(((Ad.double().abs() @ Bd.double().abs().t()) / ref.abs().clamp_min(1e-300)).median())
                report(f"x6 adversarial gemm K{K} cancel: cancellation factor >= 1e4 (median {canc:.1e})", 0.0 if canc >= 1e4 else 1.0, 0)
            if big:      # dW-shaped: C[M,N] = A^T B with the long axis contracted (TN, split-K slabs)
                At, Bt = Ad.t().contiguous(), Bd.t().contiguous()          # stored [K, M], [K, N]

                def run():
                    C = torch.empty(M, Nn, device=dev)
                    ops.gemm(1, 0, M, Nn, K, At, M, Bt, Nn, C, Nn)
                    return C
            else:
                def run():
                    C = torch.empty(M, Nn, device=dev)
                    ops.gemm(0, 1, M, Nn, K, Ad, K, Bd, K, C, Nn)
                    return C
            c32, c6 = both(run)
            grade(f"gemm {'TN' if big else 'NT'} {M}x{Nn}x{K}", tag, c32, c6, ref, scale, K, True)
            del Ad, Bd, ref
    # ---- list attention: contraction over hd (scores) and over the B lists (P V, dK, dV)
    for (B, HD) in ((256, 16), (256, 64), (2048, 16), (2048, 64)):
        S, H = 2, 2
        E = H * HD
        for tag in ("ones", "worst-split", "cancel"):
            if tag == "cancel":      # nearly uniform weights over values of alternating sign: P V cancels by ~1e4 sqrt(B)
                q = 0.01 * torch.randn(B, S, E, generator=g, device=dev)
                k = 0.01 * torch.randn(B, S, E, generator=g, device=dev)
                sign = (1.0 - 2.0 * (torch.arange(B, device=dev) % 2)).view(B, 1, 1)
                v = sign * (1.0 + 1e-4 * torch.randn(B, S, E, generator=g, device=dev))
            else:
                pat = 0xFFFF if tag == "ones" else 0x7F40
                q = _low_mantissa(0.25 * (torch.rand(B, S, E, generator=g, device=dev) + 0.5), pat)
                k = _low_mantissa(0.25 * (torch.rand(B, S, E, generator=g, device=dev) + 0.5), pat)
                v = _low_mantissa(torch.rand(B, S, E, generator=g, device=dev) + 0.5, pat)
            qkv = torch.cat([q, k, v], dim=2)
            dout = _low_mantissa(torch.rand(B, S, E, generator=g, device=dev) + 0.5, 0x7F40) if tag != "cancel" else torch.randn(B, S, E, generator=g, device=dev)
            qr = qkv.to(dev).double().requires_grad_(True)
            orf = _attn_ref(qr, H)
            orf.backward(dout.to(dev).double())

            def run():
                qd = _pm(qkv).to(dev).requires_grad_(True)
                od = ops.list_attention(qd, S, B, H)
                od.backward(_pm(dout).to(dev))
                return _unpm(od.detach(), B, S), _unpm(qd.grad, B, S)
            (o32, g32), (o6, g6) = both(run)
            vabs = float(qkv[..., 2 * E:].abs().max())
            grade(f"attn B{B} hd{HD} out", tag, o32, o6, orf.detach(), vabs, B, False)       # sum_k P_k |v_k| <= max |v|
            for nm, sl in (("dq", slice(0, E)), ("dk", slice(E, 2 * E)), ("dv", slice(2 * E, 3 * E))):
                gref = qr.grad[..., sl]
                grade(f"attn B{B} hd{HD} {nm}", tag, g32[..., sl], g6[..., sl], gref, float(gref.abs().max()), B, False)
            del qr, orf
    # the worst case over the three operand classes, per shape: the coherent pattern hits one kernel or the other
    shapes = sorted({n for n, *_ in summary}, key=[n for n, *_ in summary].index)
    for n in shapes:
        w32 = max(e32 for (m, _t, e32, _e6, _r) in summary if m == n)
        w6 = max(e6 for (m, _t, _e32, e6, _r) in summary if m == n)
        print(f"     worst over classes {n}: f32-MFMA {w32:.3e}, bf16x6 {w6:.3e}, ratio {w6 / max(w32, 1e-300):.2f}", flush=True)
        if n.startswith("gemm"):
            K = int(n.split("x")[-1])
            report(f"x6 adversarial {n}: worst class of bf16x6 <= 1.25 x worst class of the f32 MFMA + min(sqrt(K) 2^-25, 2 x it)", w6,
                   1.25 * w32 + min(math.sqrt(K) * 2.0 ** -25, 2.0 * w32))
        elif " out" in n or " dv" in n or "hd16" in n:
            # attention: the same worst-class-vs-worst-class statement - every output at head dim 16 (two accumulators), O and dV
            # at head dim 64 (dQ / dK there inherit the coherent rounding through delta = rowsum(dO o O) and dS = P (dP - delta))
            B_ = int(n.split()[1][1:])
            report(f"x6 adversarial {n}: worst class of bf16x6 <= 1.5 x worst class of the f32 MFMA + min(sqrt(B) 2^-25, 2 x it)", w6,
                   1.5 * w32 + min(math.sqrt(B_) * 2.0 ** -25, 2.0 * w32))


@section
def precision_argument():
    """The mode is a call argument (ABI v3): two encoder layers and two BiLSTM stacks run in one process in DIFFERENT modes
    (ops.precision scopes; the process default is a third one and stays untouched), each tape node keeps its forward's mode
    for its backward, and every result equals - bit for bit - the same computation with that mode as the process default."""
    from models._common import ParamTree
    keep = N.get_precision()
    try:
        B, S, E, H = 70, 5, 256, 4
        torch.manual_seed(3)
        layer = torch.nn.TransformerEncoderLayer(d_model=E, nhead=H, dropout=0.0)
        lstm = torch.nn.LSTM(3, 128, num_layers=2, batch_first=True, bidirectional=True)
        x, dy = torch.randn(S * B, E), torch.randn(S * B, E)
        x3, dh = torch.randn(S * B, 3), torch.randn(S * B, 256)

        def run_all(scope_mode):
            """forward under the scope, backward OUTSIDE it (the tape node must remember)"""
            pl, pt = ParamTree(layer).to(dev), ParamTree(lstm).to(dev)
            xin, x3in = x.clone().to(dev).requires_grad_(True), x3.clone().to(dev).requires_grad_(True)
            with ops.precision(scope_mode):
                y = ops.encoder_layer(xin, pl, S, B, H, 0.0)
                h = ops.bilstm(x3in, pt, S, B)
            y.backward(dy.to(dev))
            h.backward(dh.to(dev))
            return [y.detach(), xin.grad, h.detach(), x3in.grad] + [q.grad for _, q in pl.named_parameters()] + [q.grad for _, q in pt.named_parameters()]

        want = {}
        for mode in ("fp32", "bf16x6", "bf16x3"):
            N.set_precision(mode)
            want[mode] = run_all(None)                       # the mode as the process default
        for default in ("bf16x3", "fp32"):
            N.set_precision(default)
            for mode in ("fp32", "bf16x6", "bf16x3"):
                got = run_all(mode)
                worst = max(float((a - b).abs().max()) for a, b in zip(got, want[mode]))
                report(f"precision argument: {mode} as an argument under default {default} == {mode} as the default (max |diff|)", worst, 0)
            report(f"precision argument: the default is still {default}", 0.0 if N.get_precision() == default else 1.0, 0)
        d36 = max(float((a - b).abs().max()) for a, b in zip(want["bf16x3"], want["bf16x6"]))
        report("precision argument: the modes do differ (bf16x3 vs bf16x6 results, max |diff| > 0)", 0.0 if d36 > 0 else 1.0, 0)
    finally:
        N.set_precision(keep)


@section
def determinism():
    """Run-to-run determinism, asserted: BASELINE configs[1] (AttnCut 4096 x 300) and configs[2]
    (Choopy 8192 x 300), one full training step (forward, loss + cut metrics, backward) run TWICE from the same state in each
    precision mode: the flat gradient bucket, the cut distribution and the cut positions must be bitwise equal.  No kernel
    of the path uses float atomics; every cross-workgroup sum (split-K slabs, LayerNorm / head / bias partials, the loss
    partials) is reduced in a fixed order - this is the property those designs were paid for (attention.hip header)."""
    import gc
    import models as hm
    from utils import losses as hl
    from utils.metrics import Metric
    from rlt_hip.parallel import FlatModel
    from bench import synth_batch
    keep = N.get_precision()
    try:
        for name, B, S in (("attncut", 4096, 300), ("choopy", 8192, 300)):
            torch.manual_seed(11)
            if name == "attncut":
                model, crit, nf = hm.AttnCut(input_size=3, dropout=0.0).to(dev), hl.DivLoss(metric="f1", div_type="js", augmented=True), 3
            else:
                model, crit, nf = hm.Choopy(seq_len=S, dropout=0.0).to(dev), hl.ChoopyLoss(metric="f1"), 1
            flat = FlatModel(model)
            x, y = synth_batch(B, S, nf, 20240, dev)
            for mode in ("bf16x6", "fp32", "bf16x3"):
                N.set_precision(mode)
                runs = []
                for _ in range(2):
                    flat.zero_grad()
                    model.train()
                    out = model(x)
                    loss, k, f1, dcg = Metric.step(crit, out, y)
                    loss.backward()
                    torch.cuda.synchronize()
                    runs.append((flat.flat_grad.clone(), out.detach().clone(), k.clone(), loss.detach().clone()))
                    del out, loss
                (g0, p0, k0, l0), (g1, p1, k1, l1) = runs
                bits = lambda t: t.contiguous().view(torch.int32)
                report(f"determinism {name} {B}x{S} {mode}: flat_grad elements differing between two runs", float((bits(g0) != bits(g1)).sum()), 0)
                report(f"determinism {name} {B}x{S} {mode}: p elements differing", float((bits(p0) != bits(p1)).sum()), 0)
                report(f"determinism {name} {B}x{S} {mode}: cut positions differing", float((k0 != k1).sum()), 0)
                report(f"determinism {name} {B}x{S} {mode}: loss bits differing", float((bits(l0) != bits(l1)).sum()), 0)
                report(f"determinism {name} {B}x{S} {mode}: gradient is finite and non-zero", 0.0 if bool(torch.isfinite(g0).all()) and float(g0.abs().max()) > 0 else 1.0, 0)
                del runs, g0, g1, p0, p1
            del model, flat, x, y
            gc.collect()
            torch.cuda.empty_cache()
    finally:
        N.set_precision(keep)


@section
def _bench_ranks(world, batch, tag):
    """`bench.py --gpus 2` end to end: the parent never touches the GPU and starts its own two ranks under
    torch.distributed.run (spawn_ranks); here both ranks share this box's one GPU over gloo (RLT_DIST_BACKEND=gloo
    RLT_BENCH_DEVICE=0 - the rehearsal form; the driver's 8-GPU run uses one GPU per rank over RCCL).  Asserted: exit status,
    exactly one JSON line, n_gpus, collective.ranks, the global batch, value = all ranks' lists / the max-over-ranks time,
    no single-GPU-only blocks, a finite training state."""
    import json
    import subprocess
    bench_py = os.path.join(REPO, "bench.py")
    steps = 3
    env = dict(os.environ, RLT_DIST_BACKEND="gloo", RLT_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    res = subprocess.run([sys.executable, bench_py, "--gpus", str(world), "--batch", str(batch), "--steps", str(steps), "--warmup", "1",
                          "--other-steps", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    report(f"{tag}: exit status", float(res.returncode), 0)
    lines = [l for l in res.stdout.strip().splitlines() if l.startswith("{")]
    report(f"{tag}: exactly one JSON line on stdout", abs(len(lines) - 1), 0)
    if res.returncode or len(lines) != 1:
        print(res.stdout[-1500:], res.stderr[-3000:])
        return
    d = json.loads(lines[0])
    report(f"{tag}: n_gpus == world", abs(d["n_gpus"] - world), 0)
    report(f"{tag}: collective.ranks == world", abs((d["collective"] or {}).get("ranks", 0) - world), 0)
    report(f"{tag}: collective.backend == gloo (rehearsal)", 0.0 if (d["collective"] or {}).get("backend") == "gloo" else 1.0, 0)
    report(f"{tag}: config.global_batch == world x batch", abs(d["config"]["global_batch"] - world * batch), 0)
    report(f"{tag}: config.parallelism == dp<world>", 0.0 if d["config"]["parallelism"] == f"dp{world}" else 1.0, 0)
    report(f"{tag}: scaling == weak", 0.0 if d["scaling"] == "weak" else 1.0, 0)
    elapsed = d["ms_per_step"] * 1e-3 * steps
    report(f"{tag}: value == global batch x steps / elapsed", abs(d["value"] - world * batch * steps / elapsed) / d["value"], 1e-3)
    report(f"{tag}: hbm_kernel is None (single-GPU block)", 0.0 if d.get("hbm_kernel") is None else 1.0, 0)
    report(f"{tag}: no cpu_baseline block off rank-0-at-N=1", 0.0 if "cpu_baseline" not in d else 1.0, 0)
    vals = [d["train_state"][k] for k in ("loss", "f1", "dcg")]
    report(f"{tag}: finite training state", 0.0 if all(math.isfinite(v) for v in vals) else 1.0, 0)
    report(f"{tag}: headline precision is the library default bf16x6", 0.0 if d.get("precision_mode") == "bf16x6" else 1.0, 0)
    ev = d.get("step_hip_events") or {}
    report(f"{tag}: per-step HIP-event statistics present (median <= mean x 1.5, min <= median)",
           0.0 if ev and ev["min_ms"] <= ev["median_ms"] <= 1.5 * ev["mean_ms"] else 1.0, 0)


@section
def bench_two_ranks():
    _bench_ranks(2, 96, "bench --gpus 2")
bench_two_ranks.__doc__ = _bench_ranks.__doc__


@section
def rccl_two_ranks():
    """RCCL with MORE than one rank, once, before the driver's 8-GPU run finds out.  This pool leases one GPU per
    box, so both ranks of `bench.py --gpus 2 --batch 32` are pointed at cuda:0 (RLT_BENCH_DEVICE=0) over the real backend
    (RLT_DIST_BACKEND=nccl).  Two outcomes are acceptable: (a) RCCL runs two ranks on one device - then the N = 2 JSON line is checked
    like bench_two_ranks; (b) RCCL refuses the duplicate GPU - then the point of the test is that bench.py FAILS FAST: every rank
    exits non-zero with RCCL's own message inside the bounded wait (RLT_DIST_TIMEOUT_S), nothing hangs at a barrier (a hung rank
    is what would burn the driver's 8-GPU lease)."""
    import json
    import subprocess
    bench_py = os.path.join(REPO, "bench.py")
    env = dict(os.environ, RLT_DIST_BACKEND="nccl", RLT_BENCH_DEVICE="0", RLT_DIST_TIMEOUT_S="60", NCCL_DEBUG="WARN")
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(v, None)
    t0 = time.time()
    try:
        res = subprocess.run([sys.executable, bench_py, "--gpus", "2", "--batch", "32", "--steps", "3", "--warmup", "1", "--other-steps", "0",
                              "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=240)
        rc, out, err, hung = res.returncode, res.stdout, res.stderr, False
    except subprocess.TimeoutExpired as e:
        rc, out, err, hung = -9, (e.stdout or b"").decode(errors="replace"), (e.stderr or b"").decode(errors="replace"), True
    dt = time.time() - t0
    report("rccl_two_ranks: bench.py --gpus 2 over nccl on one device returned (no hang) within 240 s", 1.0 if hung else 0.0, 0)
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    if rc == 0:
        report("rccl_two_ranks: exactly one JSON line on stdout", abs(len(lines) - 1), 0)
        d = json.loads(lines[0]) if lines else {}
        report("rccl_two_ranks: n_gpus == 2", abs(d.get("n_gpus", 0) - 2), 0)
        report("rccl_two_ranks: collective.ranks == 2 over nccl", 0.0 if d.get("collective", {}).get("ranks") == 2 and d.get("collective", {}).get("backend") == "nccl" else 1.0, 0)
        report("rccl_two_ranks: value is finite and positive", 0.0 if d.get("value", 0) > 0 else 1.0, 0)
        print(f"   (rccl_two_ranks: RCCL ran two ranks on one device in {dt:.0f} s: {d.get('value')} lists/s, all-reduce {d.get('collective')})", flush=True)
    else:
        refused = any(m in err for m in ("Duplicate GPU", "duplicate GPU", "ncclInvalidUsage", "invalid usage", "ncclUnhandledSystemError", "NCCL error", "ncclSystemError"))
        report(f"rccl_two_ranks: non-zero exit ({rc}) carries RCCL's own refusal message", 0.0 if refused else 1.0, 0)
        report("rccl_two_ranks: refusal reached the parent inside the bounded wait (<= 180 s)", 0.0 if dt <= 180 else 1.0, 0)
        report("rccl_two_ranks: no JSON line was printed by a failed run", float(len(lines)), 0)
        tail = " | ".join(l for l in err.strip().splitlines() if "NCCL" in l or "Duplicate" in l or "Error" in l)[-400:]
        print(f"   (rccl_two_ranks: RCCL refused two ranks on one device after {dt:.0f} s, exit {rc}: {tail})", flush=True)


@section
def dp_four_ranks():
    """Many-rank rehearsal: the driver's 8-GPU run starts 8 ranks; a one-GPU box of this pool admits at
    most SIX processes on the card (the pool's process guard: a five-rank attempt was killed at 7 - this probe process and the
    launcher's parent count too), so the rehearsal runs the largest world that is allowed here - `bench.py --gpus 4 --batch 32`
    (its own torch.distributed.run child, four gloo ranks sharing this GPU: LOCAL_RANK 0..3, the MAX all-reduce of the timings
    over four ranks, global batch 128) and run.py MMOECut(4 experts, tasks 2.1) under four ranks with batches of 3 lists
    (shards 1,1,1,0 and 1,1,0,0: ragged AND empty shards every step), replicas bitwise equal.  World size 8 itself is covered on
    the CPU (tests/test_parallel_gloo.py: eight gloo ranks, gradient average against the shard-wise oracle, empty shards)."""
    from oracle import losses as ol, models as om
    _bench_ranks(4, 32, "bench --gpus 4")
    _trainer_dp_case("dp_four_ranks mmoecut(4e,2.1)", "mmoecut", ["--num-experts", "4", "--num-tasks", "2.1"],
                     lambda: om.MMOECut(seq_len=300, num_experts=4, num_tasks=2.1, input_size=3, dropout=0.0),
                     lambda: ol.MtCutLoss(metric="f1", num_tasks=2.1), "mmoecut.pkl", 29741, epochs=1, bs=3, world=4)


if __name__ == "__main__":
    want = [w for w in sys.argv[1:] if not w.startswith("--")]
    for w in sys.argv[1:]:
        if w.startswith("--precision="):
            N.set_precision(w.split("=", 1)[1])
    print("precision mode:", N.get_precision(), flush=True)
    secs = [v for v in list(globals().values()) if getattr(v, "_is_section", False)]
    for fn in secs:
        if want and fn.__name__ not in want:
            continue
        print(f"\n=== {fn.__name__} ===", flush=True)
        t_sec = time.time()
        try:
            fn()
            torch.cuda.synchronize()
            print(f"--- {fn.__name__}: {time.time() - t_sec:.1f} s", flush=True)
        except Exception:
            traceback.print_exc()
            RESULTS.append((fn.__name__ + " EXCEPTION", float("nan"), 0, False))
    bad = [r for r in RESULTS if not r[3]]
    print(f"\n{len(RESULTS) - len(bad)} ok, {len(bad)} failed")
    for r in bad:
        print("FAILED:", r[0], r[1])
    sys.exit(1 if bad else 0)
