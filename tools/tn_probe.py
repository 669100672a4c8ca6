#!/usr/bin/env python3
"""GPU box: weight-gradient products (A stored [K][M]) at Choopy's shapes, time per precision mode - which of them leave the six-product kernels."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "ranked-list-truncation_amd"))
import torch
from rlt_hip import native as N, ops
dev = torch.device("cuda")


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192 * 300
for M, Nn in ((2048, 128), (384, 128), (128, 128), (128, 2048), (512, 128)):
    A = torch.randn(T, M, device=dev)
    Bm = torch.randn(T, Nn, device=dev)
    C = torch.empty(M, Nn, device=dev)
    cs = torch.empty(M, device=dev)
    line = f"TN {M}x{Nn}x{T}:"
    for mode in ("fp32", "bf16x6"):
        N.set_precision(mode)
        ms = timeit(lambda: ops.gemm(1, 0, M, Nn, T, A, M, Bm, Nn, C, Nn, colsum_a=cs))
        line += f"  {mode} {ms:7.3f} ms {2.0 * M * Nn * T / ms / 1e9:6.1f} TF/s |"
    print(line, flush=True)
    del A, Bm
