#!/usr/bin/env python3
"""GPU box: the reference's own batch sizes (hyper_parameter_*.conf: 32 / 63 / 64 lists) are launch-bound - can the whole training
step (zero_grad + forward + loss/metrics + backward + Adam) be captured into ONE hipGraph and replayed?  Times eager steps against
graph replays of the same step and checks that the replayed step leaves the same state as an eager one."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "ranked-list-truncation_amd"))
import torch

import bench
import models as hm
from rlt_hip import native
from rlt_hip.parallel import FlatModel, FusedAdam
from utils import losses as hl
from utils.metrics import Metric

dev = torch.device("cuda")


def build(name, B):
    torch.manual_seed(1234)
    if name == "choopy":
        model, crit, nf = hm.Choopy(seq_len=300, dropout=0.0).to(dev), hl.ChoopyLoss(metric="f1"), 1
    else:
        model, crit, nf = hm.AttnCut(input_size=3, dropout=0.0).to(dev), hl.DivLoss(metric="f1", div_type="js", augmented=True), 3
    flat = FlatModel(model)
    opt = FusedAdam(flat, lr=3e-5, weight_decay=0.0014756345581373493)
    x, y = bench.synth_batch(B, 300, nf, 20240, dev)
    state = {}

    def step():
        model.train()
        opt.zero_grad()
        out = model(x)
        loss, _k, f1, dcg = Metric.step(crit, out, y)
        loss.backward()
        opt.step()
        state["loss"], state["f1"] = loss, f1
    return step, state, flat


def wall(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, B in (("choopy", 32), ("attncut", 63), ("attncut", 32)):
    step, state, flat = build(name, B)
    for _ in range(3):
        step()
    eager = wall(step, 20)
    p_eager = flat.flat_param.clone()
    l_eager = float(state["loss"])
    # a second, identically initialised replica for the graph
    step2, state2, flat2 = build(name, B)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step2()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            step2()
    except Exception as e:          # noqa: BLE001
        print(f"{name} B{B}: eager {eager:.3f} ms/step; capture FAILED: {type(e).__name__}: {str(e)[:300]}", flush=True)
        continue
    # replica 2 has done 3 warm-up steps + (capture does not execute); replay 20 -> 23 steps, like replica 1's 3 + 20
    replay = wall(g.replay, 20)
    diff = float((flat2.flat_param - p_eager).abs().max())
    print(f"{name} B{B}: eager {eager:.3f} ms/step, graph replay {replay:.3f} ms/step; after 23 steps each: max |param diff| {diff:.3e}, "
          f"loss eager {l_eager:.6e} / graph {float(state2['loss']):.6e}", flush=True)
