#!/usr/bin/env python3
"""Headline benchmark: ranked-lists/sec of the full training step (zero_grad + forward + reward loss
+ backward [+ gradient all-reduce] + Adam + cut metrics) of AttnCut on synthetic robust04-shaped
lists of length 300, batch 4096 per GPU (BASELINE.json configs[1]), on the HIP hot path.

    python bench.py [--gpus N --steps K --warmup W]          # N=1 directly
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # one rank per GPU

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the launch stream
for the dominant kernel (the attention dK/dV backward kernel); `cpu_baseline` times the CPU oracle
(oracle/, the pinned restatement of the reference) on a bounded sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "ranked-list-truncation_amd"))

import numpy as np
import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X dense fp32 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X dense bf16 MFMA peak (same guide)


def synth_batch(batch, seq_len, n_feat, seed, device):
    """robust04-shaped synthetic lists: descending N(3,2.5^2) scores, U(0,1) extra
    features, labels ~ Bernoulli(0.55 exp(-j/45) + 0.02) with at least one positive."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    scores = torch.sort(torch.randn(batch, seq_len, generator=g) * 2.5 + 3.0, dim=1, descending=True)[0]
    cols = [scores.unsqueeze(2)]
    if n_feat > 1:
        cols.append(torch.rand(batch, seq_len, n_feat - 1, generator=g))
    x = torch.cat(cols, dim=2).contiguous()
    prob = 0.55 * torch.exp(-torch.arange(seq_len, dtype=torch.float32) / 45.0) + 0.02
    y = (torch.rand(batch, seq_len, generator=g) < prob).float()
    y[y.sum(1) == 0, 0] = 1.0
    return x.to(device), y.to(device)


def cpu_baseline(seq_len, sample_batch, steps):
    """The CPU oracle's training step (fwd + vectorised reward loss + bwd) on the host cores."""
    from oracle import losses as olosses, models as omodels
    # threads = this process's CPU share (the GPU box gives 16 of the host's cores to a 1-GPU job;
    # os.cpu_count() would report the whole host and oversubscribe)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = int(os.environ.get("RLT_CPU_THREADS", min(avail, 16)))
    torch.set_num_threads(cores)
    model = omodels.AttnCut(dropout=0.0)
    crit = olosses.DivLoss(metric='f1', div_type='js', augmented=True)
    x, y = synth_batch(sample_batch, seq_len, 3, 20240, "cpu")

    def step():
        model.zero_grad()
        loss = crit(model(x), y)
        loss.backward()

    t0 = time.time()
    step()
    warm = time.time() - t0
    print(f"[bench] cpu baseline warm-up step: {warm:.1f}s on {cores} threads", file=sys.stderr, flush=True)
    steps = max(1, min(steps, int(20.0 / max(warm, 1e-3))))       # bound the CPU leg to ~20 s
    t0 = time.time()
    for i in range(steps):
        step()
        print(f"[bench] cpu baseline step {i + 1}/{steps}", file=sys.stderr, flush=True)
    dt = (time.time() - t0) / steps
    return {"value": round(sample_batch / dt, 3), "unit": "lists/s", "cores": cores, "kind": "port",
            "sample": f"oracle AttnCut+DivLoss(js,f1) fwd+bwd, batch {sample_batch} x len {seq_len}, "
                      f"{steps} steps after 1 warm-up, closed-form reward (the reference's python reward loop "
                      f"adds ~20 ms per list on top); attention cost grows with batch, so per-list CPU cost at "
                      f"batch 4096 is higher than at this sample"}


def step_algorithmic_flops(model, B, S):
    """AttnCut fwd FLOPs per token = 3,676,672 + 1024*L (L = lists the attention spans); fwd+bwd = 3x."""
    assert model == "attncut"
    return 3.0 * (3676672 + 1024 * B) * S * B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="ranked lists per GPU")
    ap.add_argument("--seq-len", type=int, default=300)
    ap.add_argument("--model", default="attncut", choices=["attncut", "choopy", "mtattncut", "mmoecut"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default=None, choices=["bf16x3", "fp32"],
                    help="MFMA product mode of the library (default: the library default, bf16x3)")
    ap.add_argument("--cpu-sample-batch", type=int, default=128)
    # side configurations of SURVEY.md 8(d); the headline line uses none of them
    ap.add_argument("--dropout", type=float, default=0.0,
                    help="train-mode dropout (parity runs use 0.0; the reference conf values are 0.4 / 0.2)")
    ap.add_argument("--num-tasks", type=float, default=None, help="mtattncut / mmoecut: 3, 2.1 or 2.2")
    ap.add_argument("--reward", default="f1", choices=["f1", "dcg"], help="reward metric of the criterion")
    ap.add_argument("--buckets", default=None,
                    help="comma-separated list lengths served round-robin, one homogeneous batch per step "
                         "(BASELINE configs[4]: 100,200,300); --steps should be a multiple of their number")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP hot path has no CPU fallback")
    # RLT_BENCH_DEVICE / RLT_DIST_BACKEND exist only to rehearse the N>1 launch contract on a one-GPU box
    # (several ranks on cuda:0 over gloo); the driver's runs use one GPU per rank over RCCL.
    dev_index = int(os.environ.get("RLT_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("RLT_DIST_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import models as hip_models
    from utils import losses as hip_losses
    from utils.metrics import Metric
    from rlt_hip import native, ops
    from rlt_hip.parallel import FlatModel, FusedAdam
    if args.precision:
        native.set_precision(args.precision)
    precision = native.get_precision()

    torch.manual_seed(1234)
    S, B = args.seq_len, args.batch
    if args.model == "attncut":
        model = hip_models.AttnCut(input_size=3, dropout=args.dropout).to(dev)
        crit = hip_losses.DivLoss(metric=args.reward, div_type='js', augmented=True)
        n_feat, heads_, hd, layers, wl = 3, 4, 64, 1, f"AttnCut + DivLoss(js,{args.reward},augmented) [BASELINE configs[1]]"
    elif args.model == "choopy":
        model = hip_models.Choopy(seq_len=S, dropout=args.dropout).to(dev)
        crit = hip_losses.ChoopyLoss(metric=args.reward)
        n_feat, heads_, hd, layers, wl = 1, 8, 16, 3, f"Choopy + ChoopyLoss({args.reward}) [BASELINE configs[2] at batch 8192]"
    elif args.model == "mtattncut":
        nt = 3 if args.num_tasks is None else args.num_tasks
        model = hip_models.MtAttnCut(input_size=3, num_tasks=nt, dropout=args.dropout).to(dev)
        crit = hip_losses.MtCutLoss(metric=args.reward, num_tasks=nt)
        n_feat, heads_, hd, layers = 3, 4, 64, 1
        wl = f"MtAttnCut(tasks {nt:g}) + MtCutLoss({args.reward}) [BASELINE configs[4]]"
    else:
        nt = 2.1 if args.num_tasks is None else args.num_tasks
        model = hip_models.MMOECut(seq_len=S, num_experts=4, num_tasks=nt, dropout=args.dropout).to(dev)
        crit = hip_losses.MtCutLoss(metric=args.reward, rerank_weight=0.4, classi_weight=0.6, num_tasks=nt)
        n_feat, heads_, hd, layers = 3, 4, 64, 4
        wl = f"MMOECut(4 experts, tasks {nt:g}) + MtCutLoss({args.reward}) [BASELINE configs[3]]"
    flat = FlatModel(model)
    flat.broadcast_params()
    opt = FusedAdam(flat, lr=3e-5, weight_decay=0.0014756345581373493)
    if args.dropout > 0:
        wl += f", dropout {args.dropout:g}"
    # inputs resident in HBM before timing; with --buckets one homogeneous batch per length, served round-robin
    lengths = [S] if not args.buckets else [int(v) for v in args.buckets.split(",")]
    if args.buckets:
        if args.model in ("choopy", "mmoecut"):
            raise SystemExit("--buckets: Choopy / MMOECut are built for one list length (seq_len)")
        wl += f", length buckets {lengths} round-robin"
    batches = [synth_batch(B, L, n_feat, 20240 + rank + 7 * i, dev) for i, L in enumerate(lengths)]
    timer = ops.KernelTimer()
    turn = [0]

    def step():
        x, y = batches[turn[0] % len(batches)]
        turn[0] += 1
        model.train()
        opt.zero_grad()
        out = model(x)
        loss = crit(out, y)
        loss.backward()
        flat.all_reduce_grads()
        opt.step()
        cut = out[-1] if isinstance(out, (list, tuple)) else out
        k, f1, dcg = Metric.evaluate(cut, y)                   # stays on device
        return loss, f1, dcg

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # every bucket is served at least once untimed (first use of a shape allocates its buffers)
    n_warm = args.warmup if len(batches) == 1 else max(args.warmup, len(batches))
    for i in range(n_warm):
        step()
        torch.cuda.synchronize()
        if rank == 0:
            print(f"[bench] warm-up step {i + 1}/{n_warm} done", file=sys.stderr, flush=True)
    fence()
    ops.KernelTimer.active = timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, f1, dcg = step()
    fence()
    elapsed = time.perf_counter() - t0
    ops.KernelTimer.active = None
    if rank == 0:
        print(f"[bench] {args.steps} timed steps: {elapsed / args.steps * 1e3:.1f} ms/step", file=sys.stderr, flush=True)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed

    if rank == 0:
        ksum = timer.summary()
        # dominant launch: the attention dK/dV backward (one rlt_list_attention_bwd_dkv call).
        #   fp32 mode  : one kernel, 4 MFMA products of 2*B*B*HD per (position, head) on the f32 MFMA
        #   bf16x3 mode: one fused kernel (S, dP, dV, dK) = 4 products, each executed as 3 bf16 MFMA products
        # `achieved` = ALGORITHMIC fp32-level FLOPs of the launch (products x 2*B*B*HD*S*H) / its HIP-event time.
        # `peak` = the dense MFMA peak of the arithmetic the launch runs in: 157.3 TF/s (f32 MFMA) in fp32 mode;
        # in bf16x3 mode every fp32 product costs 3 bf16 MFMA products, so the peak for fp32-level FLOPs is
        # 2500/3 = 833.3 TF/s (`executed_bf16_tflops` / 2500 is the same fraction).  DESIGN.md section 5.
        name = "attn_bwd_dkv"
        launches, ms = ksum.get(name, (0, float("nan")))
        S_mean = sum(lengths) / len(lengths)                   # positions per launch, averaged over the buckets
        unit_flops = 2.0 * B * B * hd * S_mean * heads_
        if precision == "fp32":
            kern, products, mult, peak = "attn_bwd_dkv_kernel<%d,2,%s>" % (hd, "true" if args.dropout > 0 else "false"), 4, 1, PEAK_F32_MFMA_TFLOPS
        else:
            kern, products, mult, peak = "attn3_bwd_dkv_kernel<%d,%s>" % (hd, "true" if args.dropout > 0 else "false"), 4, 3, PEAK_BF16_MFMA_TFLOPS
        algorithmic = products * unit_flops / (ms * 1e-3) / 1e12 if launches else float("nan")
        achieved, executed = algorithmic, algorithmic * mult
        peak = round(peak / mult, 1)
        # whole-step fractions SURVEY.md 8(d) asks for beside the kernel's: algorithmic fwd+bwd FLOPs and bytes per list
        headline = args.model == "attncut" and not args.buckets
        step_flop = step_algorithmic_flops(args.model, B, S) if headline else None
        step_bytes = 16.0e6 * B * S / 300.0 if headline else None
        # HBM traffic of that launch: PMC FETCH_SIZE/WRITE_SIZE collected in separate rocprofv3 --pmc passes of this
        # same command (tools/pmc_traffic.py -> profiles/r01_s_pmc_traffic.json, gfx950 FETCH_SIZE x2 correction
        # applied); only quoted for the exact workload and kernel it was measured on
        traffic = None
        step_traffic = None
        try:
            with open(os.path.join(REPO, "profiles", "r01_s_pmc_traffic.json")) as f:
                pmc = json.load(f)
            if (headline and args.dropout == 0 and B == 4096 and S == 300 and precision == "bf16x3"
                    and all(kern.startswith(k) for k in pmc["dominant_launch"])):
                traffic = pmc["traffic_bytes_per_launch"]
                step_traffic = pmc.get("step_traffic_bytes")
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "ranked-lists/sec (fwd+bwd) at len=300; F1@k vs CPU ref",
            "value": round(value, 2), "unit": "lists/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if precision == "fp32" else "f32 (bf16x3 split MFMA products, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"{wl}, batch {B} lists/GPU x len {S}, "
                                   f"full train step incl. Adam and cut metrics", "global_batch": B * world,
                       "seq_len": S if not args.buckets else lengths, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": kern,
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "executed_bf16_tflops": round(executed, 2) if mult > 1 else None,
                         "mfma_products_per_fp32_product": mult,
                         "launch_ms": round(ms, 3), "launches_timed": launches,
                         "other_kernels_ms": {k: round(v[1], 3) for k, v in ksum.items() if k != name},
                         "whole_step": None if step_flop is None else {
                             "algorithmic_tflop": round(step_flop / 1e12, 2),
                             "tflops": round(step_flop / (ms_per_step * 1e-3) / 1e12, 1),
                             "frac_of_f32_mfma_peak": round(step_flop / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 3),
                             "algorithmic_GB": round(step_bytes / 1e9, 1),
                             "hbm_GBps": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                             "frac_of_hbm_peak": round(step_bytes / (ms_per_step * 1e-3) / 8.0e12, 4),
                             # all kernels' FETCH_SIZE/WRITE_SIZE of one step (same PMC passes as `traffic`) / this run's time
                             "measured_traffic_GB": None if step_traffic is None else round(step_traffic / 1e9, 1),
                             "measured_hbm_GBps": None if step_traffic is None else round(step_traffic / (ms_per_step * 1e-3) / 1e9, 1),
                             "measured_frac_of_hbm_peak": None if step_traffic is None else round(step_traffic / (ms_per_step * 1e-3) / 8.0e12, 4)}},
            "train_state": {"loss": round(float(loss.detach()), 6), "f1": round(float(f1), 6), "dcg": round(float(dcg), 6)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, args.cpu_sample_batch, 3)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
