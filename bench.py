#!/usr/bin/env python3
"""Headline benchmark: ranked-lists/sec of the full training step (zero_grad + forward + reward loss
+ backward [+ gradient all-reduce] + Adam + cut metrics) of AttnCut on synthetic robust04-shaped
lists of length 300, batch 4096 per GPU (BASELINE.json configs[1]), on the HIP hot path.

    python bench.py [--gpus N --steps K --warmup W]          # N=1 directly; N>1 starts its own N ranks
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # one rank per GPU

Launched with --gpus N > 1 and no WORLD_SIZE in the environment, this process never touches the GPU:
it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child (fresh
processes, one rank per GPU over RCCL), lets rank 0 print the JSON line and exits with the child's
status.

Prints ONE JSON line (rank 0).  The top-level `value` / `ms_per_step` / `dtype` / `roofline` are the
library's DEFAULT mode, `bf16x6`: f32 operands, accumulation and storage; every product of the GEMM
family, the list attention and the BiLSTM recurrences from an EXACT three-way bf16 split of both
operands, six bf16 MFMA products per fp32 product, per-product error < 2^-23 (the reference computes
in fp32 end to end - models/AttnCut.py:8-14; VERDICT r03 accepted this mode as the reference's
precision: all 24 operand bits enter; errors against fp64 at or below the f32 MFMA kernels' on random operands and on
two of three adversarial operand classes - on operands that all share the worst-case low significand bits the GEMM family
reaches 17x the f32 kernels' error at K ~ 10^6, inside the fp32-chain bound, and list attention stays within 2x on every
class: include/rlt_hip.h, tools/gpu_probe.py x6_adversarial).  Two
sibling blocks time the same step in the same run, each with its own roofline: `f32_mfma_mode` (exact
fp32 products on `v_mfma_f32_32x32x2_f32`, peak 157.3 TFLOP/s) and `fast_mode` (bf16x3: 16 operand
bits, inside the 1e-4 parity bound, narrower than the reference - opt-in).  `--precision` moves another
mode to the top.  `roofline` is measured live with HIP events on the launch stream for the dominant
kernel (the attention dK/dV backward kernel) in training steps AFTER the timed region; `cpu_baseline`
times the CPU oracle (oracle/, the pinned restatement of the reference) on bounded samples on this
box's host cores.
"""
import argparse
import json
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "ranked-list-truncation_amd"))

import torch          # importing torch does not initialise the GPU
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X dense fp32 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X dense bf16 MFMA peak (same guide)
PEAK_HBM_BPS = 8.0e12
# HBM traffic per launch / per step is NOT measured by this script: it comes from separate rocprofv3 --pmc passes of
# this same command (tools/pmc_traffic.py; FETCH_SIZE / WRITE_SIZE, gfx950 corrections applied) committed here:
PMC_TRAFFIC_FILES = {m: tuple(os.path.join("profiles", f"{r}_pmc_traffic_{m}.json") for r in ("r06", "r05", "r04", "r03"))
                     for m in ("bf16x6", "fp32", "bf16x3")}


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


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _time_cpu_steps(step, budget_s, max_steps, tag):
    t0 = time.time()
    step()
    warm = time.time() - t0
    print(f"[bench] cpu baseline ({tag}) warm-up step: {warm:.1f}s", file=sys.stderr, flush=True)
    steps = max(1, min(max_steps, int(budget_s / max(warm, 1e-3))))
    t0 = time.time()
    for i in range(steps):
        step()
        print(f"[bench] cpu baseline ({tag}) step {i + 1}/{steps}", file=sys.stderr, flush=True)
    return (time.time() - t0) / steps, steps


def cpu_baseline(seq_len, vec_batch, loop_batch):
    """The CPU oracle's training step (zero_grad + fwd + reward loss + bwd) on the host cores, the two variants of
    SURVEY.md 8(d): `value` = closed-form (vectorised) reward at batch `vec_batch`; `loop_faithful` = the reference's
    B*S python reward loop (utils/losses.py:217-225 -> utils/metrics.py:85-101) at batch `loop_batch`."""
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

    def make_step(crit, batch, length):
        x, y = synth_batch(batch, length, 3, 20240, "cpu")

        def step():
            model.zero_grad()
            loss = crit(model(x), y)
            loss.backward()
        return step

    # the quoted configuration itself, bounded along the position axis: every position of the encoder (attention over the
    # 4096 lists, FFN, norms) and every LSTM step costs the same, so `sub_len` of the `seq_len` positions at the full batch
    # is a 1/(seq_len/sub_len) sample of one step of the benchmark workload
    full_batch, sub_len = 4096, 12
    sub_dt, sub_steps = _time_cpu_steps(
        make_step(olosses.DivLoss(metric='f1', div_type='js', augmented=True), full_batch, sub_len), 14.0, 3,
        f"vectorised reward, batch {full_batch} x {sub_len} of {seq_len} positions")
    vec_dt, vec_steps = _time_cpu_steps(
        make_step(olosses.DivLoss(metric='f1', div_type='js', augmented=True), vec_batch, seq_len), 10.0, 3,
        f"vectorised reward, batch {vec_batch}")
    loop_dt, loop_steps = _time_cpu_steps(
        make_step(olosses.DivLoss(metric='f1', div_type='js', augmented=True, loop=True), loop_batch, seq_len), 8.0, 3,
        f"loop-faithful reward, batch {loop_batch}")
    return {"value": round(full_batch / (sub_dt * seq_len / sub_len), 3), "unit": "lists/s", "cores": cores, "kind": "port",
            "extrapolated": True, "positions_sampled": sub_len, "positions_per_list": seq_len,
            "measured_step_seconds": round(sub_dt, 3),
            "cpu_model": _cpu_model(),
            "sample": f"EXTRAPOLATED from a bounded sample: oracle AttnCut+DivLoss(js,f1) fwd+bwd at the benchmark batch {full_batch} "
                      f"on {sub_len} of the {seq_len} positions per list ({sub_steps} steps after 1 warm-up, closed-form reward: "
                      f"{sub_dt:.2f} s per step MEASURED); value = {full_batch} / (that step time x {seq_len}/{sub_len}) - per-position "
                      f"cost is uniform (list-axis attention, FFN, one LSTM step), per-step constants are counted "
                      f"{seq_len // sub_len}x, so this is a LOWER bound on the CPU's speed; full-length steps measured outright "
                      f"are in full_length_small_batch / loop_faithful",
            "full_length_small_batch": {"value": round(vec_batch / vec_dt, 3), "unit": "lists/s",
                                        "sample": f"the same step at batch {vec_batch} x the full len {seq_len}, {vec_steps} steps "
                                                  f"after 1 warm-up (the attention cost per list is {full_batch // vec_batch}x "
                                                  f"smaller at this batch)"},
            "loop_faithful": {"value": round(loop_batch / loop_dt, 3), "unit": "lists/s",
                              "sample": f"same step with the reference's B*S python reward loop, batch {loop_batch} x len "
                                        f"{seq_len}, {loop_steps} steps after 1 warm-up"}}


def hbm_kernel_roofline(S, dev, lists=262144, reps=10):
    """The HBM-bound scan kernel of the path: fused reward loss (JS, F1) + d(loss)/dp + cut metrics in one
    pass, two ranked lists per wavefront (one per 32-lane half).  ALGORITHMIC bytes per list = read p and labels 8S + write
    dL/dp 4S + 24 B of results; HIP events on the launch stream around `reps` calls (two launches each: the pass and the
    final reduction).  The default 262,144 lists have a 944 MB working set, well beyond the 256 MB Infinity Cache."""
    from rlt_hip import native as N
    g = torch.Generator(device=dev).manual_seed(3)
    p = torch.softmax(torch.randn(lists, S, device=dev, generator=g), 1).contiguous()
    y = (torch.rand(lists, S, device=dev, generator=g) < 0.1).float()
    per_list, loss_out, dp = torch.empty(lists, device=dev), torch.empty(1, device=dev), torch.empty(lists, S, device=dev)
    k = torch.empty(lists, dtype=torch.int32, device=dev)
    f1, dcg = torch.empty(lists, dtype=torch.float64, device=dev), torch.empty(lists, dtype=torch.float64, device=dev)
    sums = torch.empty(2, dtype=torch.float64, device=dev)
    wsb = N.query("rlt_loss_metrics_workspace", lists)
    ws = torch.empty(wsb // 8 + 1, dtype=torch.float64, device=dev)
    from rlt_hip import ops as _ops
    tab = _ops.dcg_table(dev)

    def run():
        N.call("rlt_loss_metrics", N.ptr(p), N.ptr(y), None, lists, S, N.METRIC_F1, -1.0, N.LOSS_JS, 0.85, -1.0, N.ptr(per_list),
               N.ptr(loss_out), N.ptr(dp), N.ptr(k), N.ptr(f1), N.ptr(dcg), N.ptr(sums), N.ptr(tab), N.ptr(ws), wsb, N.stream())
    for _ in range(3):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / reps * 1e3
    nbytes = lists * (12.0 * S + 24)
    gbps = nbytes / us / 1e3
    # what this box streams with the scan's own access pattern and nothing else to do: one elementwise pass reading p and the
    # labels and writing dp (2 reads + 1 write of the same three arrays) - the practical roof beside the nominal 8 TB/s
    for _ in range(3):
        torch.add(p, y, out=dp)
    a.record()
    for _ in range(reps):
        torch.add(p, y, out=dp)
    b.record()
    torch.cuda.synchronize()
    ref_gbps = lists * 12.0 * S / (a.elapsed_time(b) / reps * 1e3) / 1e3
    return {"kernel": "reward_loss_h_kernel<3,true,true> + loss_metrics_final_kernel (rlt_loss_metrics)", "bound": "hbm",
            "lists": lists, "seq_len": S, "us_per_call": round(us, 2), "algorithmic_bytes": nbytes,
            "achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbps / 8000.0, 4),
            "stream_reference": {"op": "torch.add(p, labels, out=dp): 2 reads + 1 write of the same arrays on this box, HIP events",
                                 "GBps": round(ref_gbps, 1), "frac_of_it": round(gbps / ref_gbps, 4)}}


def hbm_kernel_both(S, dev):
    """The scan kernel at 262,144 lists (primary: the working set cannot sit in the Infinity Cache) and, nested, at 65,536
    lists (236 MB: may be partly served from the 256 MB cache; kept for comparison with the earlier rounds' lines)."""
    out = hbm_kernel_roofline(S, dev)
    small = hbm_kernel_roofline(S, dev, lists=65536, reps=30)
    out["at_65536_lists"] = {k: small[k] for k in ("lists", "us_per_call", "algorithmic_bytes", "achieved", "frac", "stream_reference")}
    out["at_65536_lists"]["note"] = "working set 236 MB < 256 MB Infinity Cache: not a clean HBM figure"
    return out


def step_algorithmic_flops(model, B, S):
    """AttnCut fwd FLOPs per token = 3,676,672 + 1024*L (L = lists the attention spans); fwd+bwd = 3x."""
    assert model == "attncut"
    return 3.0 * (3676672 + 1024 * B) * S * B


def spawn_ranks(n, argv):
    """--gpus N > 1 without a launcher: start N fresh ranks under torch.distributed.run (a CHILD process - never an
    exec - from this parent, which has not initialised the GPU) and return the child's exit status."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode        # rank 0 writes the JSON line to the inherited stdout


def small_batch_block(dev, S):
    """The reference's own batch sizes (hyper_parameter_*.conf: batch_size = 63 / 64; robust04 has 194 training lists): full training
    steps of AttnCut at 63 lists and Choopy at 32, default precision mode, median of 30 HIP-event-timed steps after 5 warm-ups.  There the
    step is the serial BiLSTM recurrences (1,200 time steps) and launch-bound encoder work, not a roofline - reported, not the headline."""
    import models as hip_models
    from utils import losses as hip_losses
    from utils.metrics import Metric
    from rlt_hip.parallel import FlatModel, FusedAdam
    out = {}
    for name, Bs in (("attncut", 63), ("choopy", 32)):
        torch.manual_seed(4321)
        if name == "attncut":
            model, crit, n_feat = hip_models.AttnCut(input_size=3, dropout=0.0).to(dev), hip_losses.DivLoss(metric="f1", div_type="js", augmented=True), 3
        else:
            model, crit, n_feat = hip_models.Choopy(seq_len=S, dropout=0.0).to(dev), hip_losses.ChoopyLoss(metric="f1"), 1
        flat = FlatModel(model)
        opt = FusedAdam(flat, lr=3e-5, weight_decay=0.0014756345581373493)
        x, y = synth_batch(Bs, S, n_feat, 777, dev)

        def step():
            model.train()
            opt.zero_grad()
            loss, _k, _f1, _dcg = Metric.step(crit, model(x), y)
            loss.backward()
            opt.step()
            return loss
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in evs:
            a.record()
            loss = step()
            b.record()
        torch.cuda.synchronize()
        per = sorted(a.elapsed_time(b) for a, b in evs)
        med = 0.5 * (per[14] + per[15])
        if not math.isfinite(float(loss.detach())):
            raise SystemExit(f"bench.py: non-finite loss in the small-batch {name} steps")
        out[f"{name}_b{Bs}"] = {"batch": Bs, "seq_len": S, "median_ms_per_step": round(med, 3), "min_ms": round(per[0], 3),
                               "lists_per_s": round(Bs / med * 1e3, 1), "steps": 30}
    out["note"] = ("the reference's configured batch sizes: full train steps incl. Adam and cut metrics, default precision mode, HIP events "
                   "per step; latency-bound (serial recurrences), not a roofline figure")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="ranked lists per GPU")
    ap.add_argument("--seq-len", type=int, default=300)
    ap.add_argument("--model", default="attncut", choices=["attncut", "choopy", "mtattncut", "mmoecut"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="bf16x6", choices=["bf16x6", "fp32", "bf16x3"],
                    help="MFMA product mode of the library for the headline figure (default bf16x6: fp32-faithful exact "
                         "three-way bf16 split, the library's default; fp32 = the f32 MFMA; bf16x3 = the opt-in fast mode)")
    ap.add_argument("--other-steps", "--fp32-steps", dest="other_steps", type=int, default=5,
                    help="steps timed in each of the two OTHER precision modes after the headline loop (0: skip)")
    ap.add_argument("--cpu-sample-batch", type=int, default=512, help="batch of the vectorised-reward CPU sample")
    ap.add_argument("--cpu-loop-batch", type=int, default=32, help="batch of the loop-faithful CPU sample")
    # side configurations of SURVEY.md 8(d); the headline line uses none of them
    ap.add_argument("--dropout", type=float, default=0.0,
                    help="train-mode dropout (parity runs use 0.0; the reference conf values are 0.4 / 0.2)")
    ap.add_argument("--num-tasks", type=float, default=None, help="mtattncut / mmoecut: 3, 2.1 or 2.2")
    ap.add_argument("--reward", default="f1", choices=["f1", "dcg"], help="reward metric of the criterion")
    ap.add_argument("--buckets", default=None,
                    help="comma-separated list lengths served round-robin, one homogeneous batch per step "
                         "(BASELINE configs[4]: 100,200,300); --steps should be a multiple of their number")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP hot path has no CPU fallback")
    # RLT_BENCH_DEVICE / RLT_DIST_BACKEND exist only to rehearse the N>1 launch contract on a one-GPU box
    # (several ranks on cuda:0 over gloo); the driver's runs use one GPU per rank over RCCL.
    # RLT_FORCE_DIST=1 (tests): initialise the process group and run every collective of the step even with ONE rank, so
    # that the RCCL code path (init with device_id, broadcast, all-reduce AVG, barrier) executes on a one-GPU box.
    dev_index = int(os.environ.get("RLT_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("RLT_DIST_BACKEND", "nccl")
    force_dist = os.environ.get("RLT_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # a rank that cannot join (RCCL refuses two ranks on one GPU, a peer died) must fail, not sit in a collective: bounded waits
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("RLT_DIST_TIMEOUT_S", "600")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
        assert dist.get_world_size() == args.gpus

    import models as hip_models
    from utils import losses as hip_losses
    from utils.metrics import Metric
    from rlt_hip import native, ops
    from rlt_hip.parallel import FlatModel, FusedAdam
    native.set_precision(args.precision)
    precision = native.get_precision()
    others = [m for m in ("bf16x6", "fp32", "bf16x3") if m != precision]
    ops.set_seed_stream(rank)             # decorrelates the dropout masks of the ranks (same torch seed everywhere)

    torch.manual_seed(1234)
    S, B = args.seq_len, args.batch
    if args.model == "attncut":
        model = hip_models.AttnCut(input_size=3, dropout=args.dropout).to(dev)
        crit = hip_losses.DivLoss(metric=args.reward, div_type='js', augmented=True)
        n_feat, heads_, hd, wl = 3, 4, 64, f"AttnCut + DivLoss(js,{args.reward},augmented) [BASELINE configs[1]]"
    elif args.model == "choopy":
        model = hip_models.Choopy(seq_len=S, dropout=args.dropout).to(dev)
        crit = hip_losses.ChoopyLoss(metric=args.reward)
        n_feat, heads_, hd, wl = 1, 8, 16, f"Choopy + ChoopyLoss({args.reward}) [BASELINE configs[2] at batch 8192]"
    elif args.model == "mtattncut":
        nt = 3 if args.num_tasks is None else args.num_tasks
        model = hip_models.MtAttnCut(input_size=3, num_tasks=nt, dropout=args.dropout).to(dev)
        crit = hip_losses.MtCutLoss(metric=args.reward, num_tasks=nt)
        n_feat, heads_, hd = 3, 4, 64
        wl = f"MtAttnCut(tasks {nt:g}) + MtCutLoss({args.reward}) [BASELINE configs[4]]"
    else:
        nt = 2.1 if args.num_tasks is None else args.num_tasks
        model = hip_models.MMOECut(seq_len=S, num_experts=4, num_tasks=nt, dropout=args.dropout).to(dev)
        crit = hip_losses.MtCutLoss(metric=args.reward, rerank_weight=0.4, classi_weight=0.6, num_tasks=nt)
        n_feat, heads_, hd = 3, 4, 64
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
    turn = [0]

    def step():
        x, y = batches[turn[0] % len(batches)]
        turn[0] += 1
        model.train()
        opt.zero_grad()
        out = model(x)
        loss, _k, f1, dcg = Metric.step(crit, out, y)          # loss + cut metrics in one kernel pass, on device
        loss.backward()
        flat.all_reduce_grads()
        opt.step()
        return loss, f1, dcg

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps):
        """n_steps steps bracketed by barrier + synchronize on both sides; (max-over-ranks seconds, last state, per-step HIP-event
        statistics).  The wall clock around the loop is the contract's number (`ms_per_step`, `value`); SURVEY.md 8(d)'s protocol
        - median of >= 20 hipEvent-timed steps - rides along: an event pair per step on the launch stream (the library launches
        on torch's current stream), median / mean / min over the steps, max over ranks of each."""
        fence()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_steps)]
        t0 = time.perf_counter()
        for i in range(n_steps):
            evs[i][0].record()
            state = step()
            evs[i][1].record()
        fence()
        elapsed = time.perf_counter() - t0
        per = sorted(a.elapsed_time(b) for a, b in evs)
        stats = [per[len(per) // 2] if len(per) % 2 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2]), sum(per) / len(per), per[0]]
        if use_dist:
            tmax = torch.tensor([elapsed] + stats, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed, stats = float(tmax[0].item()), [float(v) for v in tmax[1:]]
        return elapsed, state, {"median_ms": round(stats[0], 3), "mean_ms": round(stats[1], 3), "min_ms": round(stats[2], 3), "steps": n_steps,
                                "timer": "HIP events on the launch stream, one pair per step (max over ranks)"}

    def kernel_times(n_steps):
        """HIP-event times (on the launch stream) of the three list-attention launches of REAL training steps, taken
        OUTSIDE the timed region: with the timer active the encoder layer is driven launch by launch from the host
        (ops.EncoderLayerKernelsFn: the same launches, in the same order, as the one path-level call of the timed steps)."""
        timer = ops.KernelTimer()
        ops.KernelTimer.active = timer
        for _ in range(n_steps):
            step()
        torch.cuda.synchronize()
        ops.KernelTimer.active = None
        return timer.summary()

    def check_state(state, what):
        vals = [float(v.detach()) for v in state]
        if not all(math.isfinite(v) for v in vals):
            raise SystemExit(f"bench.py: non-finite training state after {what}: loss/f1/dcg = {vals}")
        return vals

    S_mean = sum(lengths) / len(lengths)                   # positions per launch, averaged over the buckets
    unit_flops = 2.0 * B * B * hd * S_mean * heads_        # one B x B x hd product per (position, head)
    headline = args.model == "attncut" and not args.buckets
    drop_tag = "true" if args.dropout > 0 else "false"

    def run_mode(mode, n_steps, n_warm):
        """Warm up, time n_steps (barrier + synchronize on both sides), then time the attention launches of further real
        steps with HIP events; everything in the library's precision mode `mode`."""
        native.set_precision(mode)
        # every bucket is served at least once untimed (first use of a shape allocates its buffers)
        n_warm = n_warm if len(batches) == 1 else max(n_warm, len(batches))
        for i in range(n_warm):
            step()
            torch.cuda.synchronize()
            if rank == 0:
                print(f"[bench] {mode}: warm-up step {i + 1}/{n_warm} done", file=sys.stderr, flush=True)
        elapsed, state, ev_stats = timed(n_steps)
        vals = check_state(state, f"{n_steps} timed {mode} steps")
        if rank == 0:
            print(f"[bench] {mode}: {n_steps} timed steps, {elapsed / n_steps * 1e3:.1f} ms/step", file=sys.stderr, flush=True)
        ksum = kernel_times(len(batches) * 3)
        return {"mode": mode, "steps": n_steps, "elapsed": elapsed, "ms_per_step": elapsed / n_steps * 1e3,
                "value": B * world * n_steps / elapsed, "state": vals, "ksum": ksum, "ev_stats": ev_stats}

    def roofline_block(res):
        """Dominant launch of the step: the attention dK/dV backward (one rlt_list_attention_bwd_dkv call).
          fp32 mode  : attn_bwd_dkv_kernel, 4 MFMA products (S, dP, dV, dK) of 2*B*B*HD per (position, head) on the f32 MFMA
          bf16x3 mode: attn3_bwd_dkv_kernel, the same 4 products, each executed as 3 bf16 MFMA products
        `achieved` = ALGORITHMIC fp32-level FLOPs of the launch (4 x 2*B*B*HD*S*H) / its HIP-event time.  `peak` = the dense
        MFMA peak of the arithmetic the launch runs in: 157.3 TF/s (f32 MFMA) in fp32 mode; in bf16x3 mode every fp32
        product costs 3 bf16 MFMA products, so the peak for fp32-level FLOPs is 2500/3 = 833.3 TF/s
        (`executed_bf16_tflops` / 2500 is the same fraction).  DESIGN.md section 5."""
        mode, ksum, sec = res["mode"], res["ksum"], res["ms_per_step"] * 1e-3
        launches, ms = ksum.get("attn_bwd_dkv", (0, float("nan")))
        x6_attn = mode == "bf16x6" and hd <= 64 and os.environ.get("RLT_ATTN6", "1") != "0"
        if x6_attn:                               # six bf16 MFMA products per fp32 product: peak 2500 / 6
            img_tag = "true" if os.environ.get("RLT_ATTN6_IMG", "0") not in ("", "0") else "false"     # staged from tile images?
            kern, mult, peak = f"attn6_bwd_dkv_kernel<{hd},{drop_tag},{img_tag}>", 6, PEAK_BF16_MFMA_TFLOPS
            # head dim 64: the one-wavefront-per-SIMD form, under the same condition as attn6_launch (csrc/attention6.hip): its
            # loaders form B * ld in 24-bit multiplies
            if hd == 64 and os.environ.get("RLT_A6_DKV1", "1") != "0" and B * 3 * heads_ * hd < (1 << 24):
                kern = f"attn6_bwd_dkv1_kernel<{drop_tag}>"
            if hd == 16 and os.environ.get("RLT_A6N", "1") != "0" and img_tag == "false":       # head dim 16: the 16x16x32 kernels (csrc/attention6n.hip)
                seed_tag = "true" if B >= 512 else "false"
                kern = f"attn6n_bwd_dkv_kernel<2,{drop_tag},{seed_tag}>"
                if args.dropout == 0 and B >= 512 and os.environ.get("RLT_A6N_1", "1") != "0":     # the pipelined one-wavefront kernel
                    kern = "attn6n_bwd1_kernel<true>"
        elif mode in ("fp32", "bf16x6"):          # (bf16x6 with RLT_ATTN6=0: the exact-fp32 kernels)
            kern, mult, peak = f"attn_bwd_dkv_kernel<{hd},2,{drop_tag}>", 1, PEAK_F32_MFMA_TFLOPS
        else:
            kern, mult, peak = f"attn3_bwd_dkv_kernel<{hd},{drop_tag}>", 3, PEAK_BF16_MFMA_TFLOPS
        achieved = 4 * unit_flops / (ms * 1e-3) / 1e12 if launches else float("nan")
        peak = round(peak / mult, 1)
        # whole-step fractions SURVEY.md 8(d) asks for beside the kernel's: algorithmic fwd+bwd FLOPs and bytes per list
        step_flop = step_algorithmic_flops(args.model, B, S) if headline else None
        step_bytes = 16.0e6 * B * S / 300.0 if headline else None
        traffic = step_traffic = traffic_source = None
        for cand in PMC_TRAFFIC_FILES[mode]:
            try:
                with open(os.path.join(REPO, cand)) as f:
                    pmc = json.load(f)
                if (headline and args.dropout == 0 and B == 4096 and S == 300
                        and all(kern.startswith(k) for k in pmc["dominant_launch"])):
                    traffic = pmc["traffic_bytes_per_launch"]
                    step_traffic = pmc.get("step_traffic_bytes")
                    traffic_source = cand + " (separate rocprofv3 --pmc passes of this command, committed; NOT measured by this run)"
                break
            except (OSError, KeyError, ValueError):
                continue
        return {"bound": "mfma", "kernel": kern,
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                "executed_bf16_tflops": round(achieved * mult, 2) if mult > 1 else None,
                "mfma_products_per_fp32_product": mult,
                "launch_ms": round(ms, 3), "launches_timed": launches,
                "timed_in": "separate training steps after the timed region, encoder layer driven launch by launch (HIP events on the launch stream)",
                "other_kernels_ms": {k: round(v[1], 3) for k, v in ksum.items() if k != "attn_bwd_dkv"},
                "whole_step": None if step_flop is None else {
                    "algorithmic_tflop": round(step_flop / 1e12, 2),
                    "tflops": round(step_flop / sec / 1e12, 1),
                    "frac_of_f32_mfma_peak": round(step_flop / sec / 1e12 / PEAK_F32_MFMA_TFLOPS, 3),
                    "frac_of_mode_mfma_peak": round(step_flop / sec / 1e12 / peak, 3),       # peak of THIS mode's products (the `peak` above)
                    "algorithmic_GB": round(step_bytes / 1e9, 1),
                    "hbm_GBps": round(step_bytes / sec / 1e9, 1),
                    "frac_of_hbm_peak": round(step_bytes / sec / PEAK_HBM_BPS, 4),
                    # all kernels' FETCH_SIZE/WRITE_SIZE of one step (same PMC file as `traffic`) / this run's time
                    "pmc_file_traffic_GB": None if step_traffic is None else round(step_traffic / 1e9, 1),
                    "pmc_file_hbm_GBps": None if step_traffic is None else round(step_traffic / sec / 1e9, 1),
                    "pmc_file_frac_of_hbm_peak": None if step_traffic is None else round(step_traffic / sec / PEAK_HBM_BPS, 4)}}

    DTYPES = {"fp32": "f32 (operands, accumulation, storage); exact fp32 products on the f32 MFMA",
              "bf16x3": "f32 storage and accumulation; MFMA products split into 3 bf16 products (hi*hi + hi*lo + lo*hi, ~16 operand mantissa bits)",
              "bf16x6": "f32 (operands, accumulation, storage); products by exact 3xbf16 split, 6 MFMA products, per-product error < 2^-23"}
    BLOCK = {"fp32": "f32_mfma_mode", "bf16x3": "fast_mode", "bf16x6": "fp32_faithful_mode"}
    NOTES = {"fp32": "the library's exact-fp32 MFMA mode: v_mfma_f32_32x32x2_f32 / 16x16x4 products, bitwise an fp32 fma chain; priced "
                     "against the 157.3 TFLOP/s f32 MFMA peak",
             "bf16x3": "the library's opt-in split-bf16 product mode: inside the 1e-4 parity bound of BASELINE.json (GPU suite green in "
                       "this mode), but its products carry 16 operand bits where the reference's fp32 carries 24 - never the headline",
             "bf16x6": "the library's default: GEMM family (csrc/gemm.hip gemm6*_kernel), list attention (csrc/attention6.hip, attention6n.hip) "
                       "and both BiLSTM recurrences (csrc/lstm6w.hip) on the exact three-way bf16 split, six MFMA products per fp32 product"}
    main_res = run_mode(precision, args.steps, args.warmup)
    other_res = [run_mode(m, args.other_steps, 1) for m in others] if args.other_steps > 0 else []
    native.set_precision(precision)

    collective = None
    if use_dist:
        # a one-rank AVG must leave the bucket unchanged (bitwise); with more ranks this is just one more all-reduce
        g0 = flat.flat_grad.clone()
        flat.all_reduce_grads()
        torch.cuda.synchronize()
        one_rank_diff = float((flat.flat_grad - g0).abs().max()) if world == 1 else None
        collective = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                      "rccl_ranks": world if dist.get_backend() == "nccl" else 0, "forced_with_one_rank": force_dist and world == 1,
                      "one_rank_avg_max_abs_diff": one_rank_diff,
                      "per_step": f"one all-reduce(AVG) of the flat fp32 gradient bucket, {flat.numel * 4 / 1e6:.2f} MB, issued after "
                                  f"the whole backward (not overlapped: the bucket is 7-27 MB, ~0.1 ms on xGMI against a >100 ms step)"}

    if rank == 0:
        loss_v, f1_v, dcg_v = main_res["state"]
        out = {
            "metric": "ranked-lists/sec (fwd+bwd) at len=300; F1@k vs CPU ref",
            "value": round(main_res["value"], 2), "unit": "lists/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(main_res["ms_per_step"], 3), "step_hip_events": main_res["ev_stats"],
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPES[precision], "precision_mode": precision,
            "data": "synthetic",
            "config": {"workload": f"{wl}, batch {B} lists/GPU x len {S}, "
                                   f"full train step incl. Adam and cut metrics", "global_batch": B * world,
                       "seq_len": S if not args.buckets else lengths, "parallelism": f"dp{world}"},
            "collective": collective,
            "roofline": roofline_block(main_res),
            "hbm_kernel": hbm_kernel_both(S, dev) if world == 1 else None,
            "train_state": {"loss": round(loss_v, 6), "f1": round(f1_v, 6), "dcg": round(dcg_v, 6)},
        }
        for res in other_res:
            m = res["mode"]
            out[BLOCK[m]] = {
                "dtype": DTYPES[m], "precision_mode": m, "steps": res["steps"],
                "ms_per_step": round(res["ms_per_step"], 3), "step_hip_events": res["ev_stats"],
                "value": round(res["value"], 2), "unit": "lists/s",
                "roofline": roofline_block(res),
                "train_state": dict(zip(("loss", "f1", "dcg"), (round(v, 6) for v in res["state"]))),
                "note": NOTES[m]}
        if world == 1 and headline and args.dropout == 0 and os.environ.get("RLT_BENCH_SMALL", "1") != "0":
            out["small_batch"] = small_batch_block(dev, S)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, args.cpu_sample_batch, args.cpu_loop_batch)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
