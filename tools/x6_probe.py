#!/usr/bin/env python3
"""GPU box: the three product modes of the GEMM family side by side - time and error against fp64 - on the products of the
4096 x 300 AttnCut step.  `python tools/x6_probe.py [T]`.  Error = max |C - C64| / max |C64| over a sample of rows (NT / NN)
or the whole product (TN), the figure the parity tests bound (tools/gpu_probe.py: mfma_tol)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "ranked-list-truncation_amd"))
import torch
from rlt_hip import native as N, ops

dev = torch.device("cuda")


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    modes = ("fp32", "bf16x3", "bf16x6")
    for a in sys.argv[1:]:
        if a.startswith("--modes="):
            modes = tuple(a.split("=", 1)[1].split(","))
    T = int(pos[0]) if pos else 4096 * 300
    shapes = [("in_proj fwd NT", 0, 1, T, 768, 256), ("out_proj fwd NT", 0, 1, T, 256, 256), ("ffn1 fwd NT", 0, 1, T, 2048, 256),
              ("ffn2 fwd NT", 0, 1, T, 256, 2048), ("ffn1 dX NN", 0, 0, T, 256, 2048), ("ffn2 dX NN", 0, 0, T, 2048, 256),
              ("ffn1 dW TN", 1, 0, 2048, 256, T), ("ffn2 dW TN", 1, 0, 256, 2048, T), ("in_proj dW TN", 1, 0, 768, 256, T),
              ("lstm dWhh TN", 1, 0, 512, 128, T), ("lstm in NT", 0, 1, T, 1024, 256), ("lstm dX NN", 0, 0, T, 256, 1024),
              ("in_proj dX NN", 0, 0, T, 256, 768), ("lstm dWih TN", 1, 0, 1024, 256, T)]
    only = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]
    if only:
        shapes = [sh for sh in shapes if any(o in sh[0] for o in only[0].split(","))]
    g = torch.Generator(device=dev).manual_seed(1)
    tot = {m: 0.0 for m in modes}
    for name, ta, tb, M, Nn, K in shapes:
        A = torch.randn((K, M) if ta else (M, K), device=dev, generator=g)
        Bm = torch.randn((Nn, K) if tb else (K, Nn), device=dev, generator=g) / (K ** 0.5 if K < 10000 else 1.0)
        bias = torch.randn(Nn, device=dev, generator=g)
        C = torch.empty(M, Nn, device=dev)
        cs = torch.empty(M, device=dev) if ta else None
        if ta:                                           # whole product in fp64, chunked over K
            ref = torch.zeros(M, Nn, dtype=torch.float64, device=dev)
            for lo in range(0, K, 131072):
                ref += A[lo:lo + 131072].double().t() @ Bm[lo:lo + 131072].double()
            ref += bias.double()
            rows = slice(None)
        else:                                            # a sample of rows
            rows = torch.randint(0, M, (4096,), device=dev, generator=g)
            ref = A[rows].double() @ (Bm.double().t() if tb else Bm.double()) + bias.double()
        line = f"{name:16s} {M}x{Nn}x{K}:"
        for mode in modes:
            N.set_precision(mode)
            ms = timeit(lambda: ops.gemm(ta, tb, M, Nn, K, A, A.shape[1], Bm, Bm.shape[1], C, Nn, bias=bias, colsum_a=cs))
            err = float((C[rows].double() - ref).abs().max() / ref.abs().max())
            tot[mode] += ms
            line += f"  {mode} {ms:7.3f} ms {2.0 * M * Nn * K / ms / 1e9:6.1f} TF/s err {err:.2e} |"
        print(line, flush=True)
        del A, Bm, C, ref
    print("sum of the products above: " + ", ".join(f"{m} {v:.2f} ms" for m, v in tot.items()))
    N.set_precision("bf16x6")


if __name__ == "__main__":
    main()
