#!/usr/bin/env python3
"""Developer probe (GPU box): time every GEMM shape of the AttnCut 4096 x 300 training step (and Choopy's 8192 x 300 with --choopy) in
the process's precision mode, one line per product: kernel family is whatever the dispatch picks under the RLT_* switches in the
environment (compare two runs).  python tools/gemm_shapes_probe.py [--choopy]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "ranked-list-truncation_amd"))
import torch
from rlt_hip import native as N
from rlt_hip import ops

dev = torch.device("cuda")


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    choopy = "--choopy" in sys.argv
    T = 8192 * 300 if choopy else 4096 * 300
    E, FF = (128, 2048) if choopy else (256, 2048)
    shapes = [  # name, ta, tb, M, N, K, accumulate
        ("in_proj fwd NT", 0, 1, T, 3 * E, E, 0), ("out_proj fwd NT", 0, 1, T, E, E, 0), ("linear1 fwd NT", 0, 1, T, FF, E, 0),
        ("linear2 fwd NT", 0, 1, T, E, FF, 0), ("linear2 dX NN", 0, 0, T, FF, E, 0), ("linear1 dX NN acc", 0, 0, T, E, FF, 1),
        ("out_proj dX NN", 0, 0, T, E, E, 0), ("in_proj dX NN acc", 0, 0, T, E, 3 * E, 1),
        ("linear2 dW TN", 1, 0, E, FF, T, 0), ("linear1 dW TN", 1, 0, FF, E, T, 0), ("out_proj dW TN", 1, 0, E, E, T, 0),
        ("in_proj dW TN", 1, 0, 3 * E, E, T, 0)]
    if not choopy:
        shapes += [("lstm1 x-proj NT", 0, 1, T, 1024, 256, 0), ("lstm1 dX NN", 0, 0, T, 256, 1024, 0), ("lstm1 dW_ih TN", 1, 0, 1024, 256, T, 0),
                   ("lstm dW_hh TN", 1, 0, 512, 128, T, 0)]
    print("env:", {k: v for k, v in os.environ.items() if k.startswith("RLT_")}, "mode", N.get_precision(), flush=True)
    tot = 0.0
    for name, ta, tb, M, Nn, K, acc in shapes:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        Bm = torch.randn((Nn, K) if tb else (K, Nn), device=dev)
        C = torch.zeros(M, Nn, device=dev)
        bias = torch.randn(Nn, device=dev) if not ta and not acc else None
        cs = torch.empty(M, device=dev) if ta else None
        fl = N.GEMM_ACCUMULATE if acc else 0
        ms = timeit(lambda: ops.gemm(ta, tb, M, Nn, K, A, A.shape[1], Bm, Bm.shape[1], C, Nn, bias=bias, flags=fl, colsum_a=cs))
        tot += ms
        print(f"{name:20s} {M:8d} x {Nn:5d} x {K:8d}: {ms:8.3f} ms  {2.0 * M * Nn * K / ms / 1e9:7.1f} TF/s", flush=True)
        del A, Bm, C
    print(f"sum {tot:.2f} ms")


if __name__ == "__main__":
    main()
