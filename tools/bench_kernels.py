#!/usr/bin/env python3
"""Developer micro-benchmark (GPU box): time the MFMA kernels at the BASELINE configs[1] shapes
(AttnCut, 4096 lists x 300) with HIP events and print TFLOP/s.  Variants are selected with the
RLT_* environment variables the library reads at launch."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "ranked-list-truncation_amd"))

import torch

from rlt_hip import native as N
from rlt_hip import ops
from rlt_hip.native import call, ptr, stream

dev = torch.device("cuda")


def timeit(fn, reps=3, warm=1):
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


def attention(B=4096, S=60, H=4, HD=64, drop=0.0):
    E = H * HD
    T = S * B
    qkv = torch.randn(T, 3 * E, device=dev)
    out = torch.empty(T, E, device=dev)
    lse = torch.empty(S, H, B, device=dev)
    dout = torch.randn(T, E, device=dev)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(S, H, B, device=dev)
    unit = B * B * HD * S * H / 1e9     # GFLOP per "2*B*B*HD" product /2
    ib = N.query("rlt_list_attention_fwd_workspace", S, B, H, HD, drop, N.PRECISION_DEFAULT)
    images = torch.empty(max(ib, 16) // 4, device=dev) if ib else None
    wb = N.query("rlt_list_attention_bwd_workspace", S, B, H, HD, drop, N.PRECISION_DEFAULT)
    ws = torch.empty(wb // 4 + 4, device=dev)
    ms = timeit(lambda: call("rlt_list_attention_fwd", ptr(qkv), S, B, H, HD, drop, 7, ptr(out), ptr(lse), ptr(images), ib, N.PRECISION_DEFAULT, stream()))
    print(f"attn_fwd      B{B} S{S} HD{HD}: {ms:8.3f} ms  {4 * unit / ms:7.1f} TF/s", flush=True)
    ms = timeit(lambda: call("rlt_list_attention_bwd_prepare", ptr(out), ptr(dout), ptr(lse), S, B, H, HD, drop, ptr(images), ptr(ws), wb, N.PRECISION_DEFAULT, stream()))
    print(f"attn_bwd_prep B{B} S{S} HD{HD}: {ms:8.3f} ms", flush=True)
    ms = timeit(lambda: call("rlt_list_attention_bwd_dkv", ptr(qkv), ptr(dout), ptr(lse), ptr(images), ptr(ws), wb, S, B, H, HD, drop, 7, ptr(dqkv), N.PRECISION_DEFAULT, stream()))
    print(f"attn_bwd_dkv  B{B} S{S} HD{HD}: {ms:8.3f} ms  {8 * unit / ms:7.1f} TF/s", flush=True)
    ms = timeit(lambda: call("rlt_list_attention_bwd_dq", ptr(qkv), ptr(dout), ptr(lse), ptr(images), ptr(ws), wb, S, B, H, HD, drop, 7, ptr(dqkv), N.PRECISION_DEFAULT, stream()))
    print(f"attn_bwd_dq   B{B} S{S} HD{HD}: {ms:8.3f} ms  {6 * unit / ms:7.1f} TF/s", flush=True)


def attention_fwd(B=4096, S=60, H=4, HD=64):
    """The forward launch alone (its prepare passes and fix-up launch included)."""
    E = H * HD
    T = S * B
    qkv = torch.randn(T, 3 * E, device=dev)
    out = torch.empty(T, E, device=dev)
    lse = torch.empty(S, H, B, device=dev)
    ib = N.query("rlt_list_attention_fwd_workspace", S, B, H, HD, 0.0, N.PRECISION_DEFAULT)
    images = torch.empty(max(ib, 16) // 4, device=dev) if ib else None
    ms = timeit(lambda: call("rlt_list_attention_fwd", ptr(qkv), S, B, H, HD, 0.0, 7, ptr(out), ptr(lse), ptr(images), ib, N.PRECISION_DEFAULT, stream()), reps=5, warm=2)
    print(f"attn_fwd      B{B} S{S} HD{HD}: {ms:8.3f} ms  {4 * B * B * HD * S * H / 1e9 / ms:7.1f} TF/s", flush=True)


def attention_drop():
    """AttnCut's conf dropout (0.4) and Choopy's (0.2)."""
    print("dropout 0.4:", flush=True)
    attention(drop=0.4)
    print("dropout 0.2, head dim 16:", flush=True)
    attention(B=8192, S=20, H=8, HD=16, drop=0.2)


def attention16():
    """Choopy's shape (BASELINE configs[2]): 8192 lists, 8 heads x 16."""
    attention(B=8192, S=20, H=8, HD=16)


def gemms(T=4096 * 300):
    shapes = [("in_proj fwd NT", 0, 1, T, 768, 256), ("ffn1 fwd NT", 0, 1, T, 2048, 256), ("ffn2 fwd NT", 0, 1, T, 256, 2048),
              ("ffn1 dX NN", 0, 0, T, 256, 2048), ("ffn2 dX NN", 0, 0, T, 2048, 256),
              ("ffn1 dW TN", 1, 0, 2048, 256, T), ("ffn2 dW TN", 1, 0, 256, 2048, T), ("in_proj dW TN", 1, 0, 768, 256, T),
              ("lstm dWhh TN", 1, 0, 512, 128, T)]
    for name, ta, tb, M, Nn, K in shapes:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        Bm = torch.randn((Nn, K) if tb else (K, Nn), device=dev)
        C = torch.empty(M, Nn, device=dev)
        bias = torch.randn(Nn, device=dev)
        ms = timeit(lambda: ops.gemm(ta, tb, M, Nn, K, A, A.shape[1], Bm, Bm.shape[1], C, Nn, bias=bias))
        print(f"gemm {name:16s} {M}x{Nn}x{K}: {ms:8.3f} ms  {2.0 * M * Nn * K / ms / 1e9:7.1f} TF/s", flush=True)
        del A, Bm, C


def gemm_bits(T=4096 * 300, E=256, Fh=2048):
    """The FFN pair with the 1-bit mask: linear1 + ReLU forward (writes hid + mask) and the masked dH backward."""
    x = torch.randn(T, E, device=dev)
    w1 = torch.randn(Fh, E, device=dev) / 16
    b1 = torch.randn(Fh, device=dev) / 10
    hid = torch.empty(T, Fh, device=dev)
    bits = ops.alloc_relu_bits(T, Fh, dev)
    ms = timeit(lambda: ops.gemm_bits(0, 1, T, Fh, E, x, E, w1, E, hid, Fh, bias=b1, flags=N.GEMM_RELU, bits_out=bits))
    print(f"gemm_bits ffn1 fwd NT  {T}x{Fh}x{E}: {ms:8.3f} ms  {2.0 * T * Fh * E / ms / 1e9:7.1f} TF/s", flush=True)
    dy = torch.randn(T, E, device=dev)
    w2 = torch.randn(E, Fh, device=dev) / 16
    ms = timeit(lambda: ops.gemm_bits(0, 0, T, Fh, E, dy, E, w2, Fh, hid, Fh, bits_in=bits))
    print(f"gemm_bits dhid bwd NN  {T}x{Fh}x{E}: {ms:8.3f} ms  {2.0 * T * Fh * E / ms / 1e9:7.1f} TF/s", flush=True)
    ms = timeit(lambda: ops.gemm(0, 1, T, Fh, E, x, E, w1, E, hid, Fh, bias=b1, flags=N.GEMM_RELU))
    print(f"gemm      ffn1 relu NT {T}x{Fh}x{E}: {ms:8.3f} ms  {2.0 * T * Fh * E / ms / 1e9:7.1f} TF/s", flush=True)
    ms = timeit(lambda: ops.gemm(0, 1, T, Fh, E, x, E, w1, E, hid, Fh))
    print(f"gemm      ffn1 plain NT {T}x{Fh}x{E}: {ms:8.3f} ms  {2.0 * T * Fh * E / ms / 1e9:7.1f} TF/s", flush=True)


def loss(S=300):
    """HBM-bound scan kernels: the fused reward-loss + cut-metrics pass (rlt_loss_metrics) against the separate kernels
    it replaces; algorithmic bytes per list = read p, labels 8S + write dL/dp 4S + 24 B of results.
    HIP-event times include the host's launch gaps at small batches: read the kernel durations from
    `rocprofv3 --kernel-trace --stats -- python tools/bench_kernels.py loss`."""
    for B in (4096, 65536, 262144):
        g = torch.Generator(device=dev).manual_seed(B)
        p = torch.softmax(torch.randn(B, S, device=dev, generator=g), 1).contiguous()
        y = (torch.rand(B, S, device=dev, generator=g) < 0.1).float()
        per_list, loss_out, dp = torch.empty(B, device=dev), torch.empty(1, device=dev), torch.empty(B, S, device=dev)
        k = torch.empty(B, dtype=torch.int32, device=dev)
        f1, dcg = torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.float64, device=dev)
        sums = torch.empty(2, dtype=torch.float64, device=dev)
        wsb = N.query("rlt_loss_metrics_workspace", B)
        ws = torch.empty(wsb // 8 + 1, dtype=torch.float64, device=dev)
        nbytes = B * (12.0 * S + 24)

        def fused():
            call("rlt_loss_metrics", ptr(p), ptr(y), None, B, S, N.METRIC_F1, -1.0, N.LOSS_JS, 0.85, -1.0, ptr(per_list), ptr(loss_out),
                 ptr(dp), ptr(k), ptr(f1), ptr(dcg), ptr(sums), ptr(ops.dcg_table(dev)), ptr(ws), wsb, stream())

        def separate():
            call("rlt_reward_loss", ptr(p), ptr(y), None, B, S, N.METRIC_F1, N.LOSS_JS, 0.85, ptr(per_list), ptr(loss_out), ptr(dp), stream())
            call("rlt_cut_metrics", ptr(p), ptr(y), None, B, S, ptr(k), ptr(f1), ptr(dcg), ptr(sums), stream())
        ms = timeit(fused, reps=50, warm=5)
        print(f"loss+metrics fused (2 launches) B{B} S{S}: {ms * 1e3:9.1f} us  {nbytes / ms / 1e6:8.1f} GB/s algorithmic = {nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
        ms = timeit(separate, reps=50, warm=5)
        print(f"loss, then metrics (4 launches) B{B} S{S}: {ms * 1e3:9.1f} us  {(nbytes + B * 8.0 * S) / ms / 1e6:8.1f} GB/s of its own bytes (p, labels read twice)", flush=True)


def lstm(B=4096, S=300):
    T = S * B
    gates = torch.randn(T, 1024, device=dev) * 0.5
    w = torch.randn(2, 512, 128, device=dev) / 12
    h = torch.empty(T, 256, device=dev)
    c = torch.empty(T, 256, device=dev)
    dh = torch.randn(T, 256, device=dev)
    fl = 2.0 * 512 * 128 * T * 2 / 1e9
    ms = timeit(lambda: call("rlt_bilstm_rec_fwd", ptr(gates), ptr(w[0]), ptr(w[1]), S, B, ptr(h), ptr(c), N.PRECISION_DEFAULT, stream()), reps=2)
    print(f"bilstm_fwd B{B} S{S}: {ms:8.3f} ms  {fl / ms:7.1f} TF/s", flush=True)
    ms = timeit(lambda: call("rlt_bilstm_rec_bwd", ptr(gates), ptr(c), ptr(w[0]), ptr(w[1]), ptr(dh), S, B, N.PRECISION_DEFAULT, stream()), reps=2)
    print(f"bilstm_bwd B{B} S{S}: {ms:8.3f} ms  {fl / ms:7.1f} TF/s", flush=True)


def lstm_x(B=4096, S=300):
    """Layer 0: the recurrence with the fused input projection (I = 3)."""
    T = S * B
    x = torch.randn(T, 3, device=dev)
    gates = torch.empty(T, 1024, device=dev)
    w = torch.randn(2, 512, 128, device=dev) / 12
    wi = torch.randn(2, 512, 3, device=dev) / 2
    bi = torch.randn(2, 512, device=dev) / 4
    bh = torch.randn(2, 512, device=dev) / 4
    h = torch.empty(T, 256, device=dev)
    c = torch.empty(T, 256, device=dev)
    ms = timeit(lambda: call("rlt_bilstm_rec_fwd_x", ptr(x), 3, ptr(wi[0]), ptr(bi[0]), ptr(bh[0]), ptr(wi[1]), ptr(bi[1]), ptr(bh[1]),
                             ptr(w[0]), ptr(w[1]), S, B, ptr(gates), ptr(h), ptr(c), N.PRECISION_DEFAULT, stream()), reps=2)
    print(f"bilstm_fwd_x B{B} S{S}: {ms:8.3f} ms", flush=True)


def lstm_w():
    """Forward recurrences (layer 1 form and layer 0 with the fused input projection) over batch sizes."""
    for B in (32, 63, 512, 4096):
        S = 300
        T = S * B
        gates = torch.randn(T, 1024, device=dev) * 0.5
        w = torch.randn(2, 512, 128, device=dev) / 12
        h = torch.empty(T, 256, device=dev)
        c = torch.empty(T, 256, device=dev)
        ms = timeit(lambda: call("rlt_bilstm_rec_fwd", ptr(gates), ptr(w[0]), ptr(w[1]), S, B, ptr(h), ptr(c), N.PRECISION_DEFAULT, stream()), reps=3)
        print(f"bilstm_fwd   B{B} S{S}: {ms:8.3f} ms  {ms / S * 1e3:6.2f} us / step", flush=True)
        lstm_x(B, S)
        dh = torch.randn(T, 256, device=dev)
        ms = timeit(lambda: call("rlt_bilstm_rec_bwd", ptr(gates), ptr(c), ptr(w[0]), ptr(w[1]), ptr(dh), S, B, N.PRECISION_DEFAULT, stream()), reps=3)
        print(f"bilstm_bwd   B{B} S{S}: {ms:8.3f} ms  {ms / S * 1e3:6.2f} us / step", flush=True)


def lstm_small():
    for B in (32, 64, 256):
        lstm(B=B)


def overlap(B=4096, S=300):
    """Does a latency/HBM-bound persistent kernel (BiLSTM backward recurrence, 256 workgroups) share the chip with an
    MFMA-bound weight-gradient product (TN, M=2048 N=256 K=T, 256 workgroups) issued on a second stream?"""
    T = S * B
    gates = torch.randn(T, 1024, device=dev) * 0.5
    w = torch.randn(2, 512, 128, device=dev) / 12
    h = torch.empty(T, 256, device=dev)
    c = torch.empty(T, 256, device=dev)
    dh = torch.randn(T, 256, device=dev)
    call("rlt_bilstm_rec_fwd", ptr(gates), ptr(w[0]), ptr(w[1]), S, B, ptr(h), ptr(c), N.PRECISION_DEFAULT, stream())
    hid = torch.randn(T, 2048, device=dev)
    dy = torch.randn(T, 256, device=dev)
    dw = torch.empty(2048, 256, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def rec():
        call("rlt_bilstm_rec_bwd", ptr(gates), ptr(c), ptr(w[0]), ptr(w[1]), ptr(dh), S, B, N.PRECISION_DEFAULT, stream())

    def gemm_tn():
        ops.gemm(1, 0, 2048, 256, T, hid, 2048, dy, 256, dw, 256)

    def both():
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(s1):
            s1.wait_event(ev)
            rec()
            e1 = torch.cuda.Event(); e1.record()
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            gemm_tn()
            gemm_tn()
            e2 = torch.cuda.Event(); e2.record()
        torch.cuda.current_stream().wait_event(e1)
        torch.cuda.current_stream().wait_event(e2)

    print(f"bilstm_bwd alone      : {timeit(rec, reps=3):8.3f} ms", flush=True)
    print(f"2 x dW (TN) alone     : {timeit(lambda: (gemm_tn(), gemm_tn()), reps=3):8.3f} ms", flush=True)
    print(f"both, two streams     : {timeit(both, reps=3):8.3f} ms", flush=True)


def stamps():
    """Timeline of one dK+dV workgroup (library built with -DRLT_STAMPS): per wavefront and tile, cycles spent computing
    (tile start -> before the barrier) and the start offsets relative to wavefront 0."""
    import ctypes
    attention()
    buf = (ctypes.c_ulonglong * 256)()
    fn = N.load().rlt_debug_stamps
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    rc = fn(buf, 256)
    assert rc == 0, rc
    v = list(buf)
    t00 = v[0]
    print("wave: per tile (start - wave0 tile0 start, compute cycles)")
    for w in range(8):
        row = []
        for t in range(2, 10):
            a, b = v[(w * 16 + t) * 2], v[(w * 16 + t) * 2 + 1]
            row.append(f"{a - t00:7d}+{b - a:5d}")
        print(f"  w{w}: " + "  ".join(row))


def gemm_stamps(T=4096 * 300):
    """Timeline of one gemm3b workgroup (library built with -DRLT_STAMPS): prologue / K loop / epilogue cycles per wavefront."""
    import ctypes
    fn = N.load().rlt_debug_gemm_stamps
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    for name, ta, tb, M, Nn, K in [("ffn1 fwd NT", 0, 1, T, 2048, 256), ("ffn2 fwd NT", 0, 1, T, 256, 2048), ("ffn2 dX NN", 0, 0, T, 2048, 256)]:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        Bm = torch.randn((Nn, K) if tb else (K, Nn), device=dev)
        C = torch.empty(M, Nn, device=dev)
        ms = timeit(lambda: ops.gemm(ta, tb, M, Nn, K, A, A.shape[1], Bm, Bm.shape[1], C, Nn))
        buf = (ctypes.c_ulonglong * 32)()
        assert fn(buf, 32) == 0
        v = list(buf)
        print(f"gemm {name} {M}x{Nn}x{K}: {ms:.3f} ms; per wavefront (prologue, K loop, epilogue) cycles:")
        for w in range(8):
            s0, s1, s2, s3 = v[4 * w:4 * w + 4]
            print(f"   w{w}: {s1 - s0:6d} {s2 - s1:7d} {s3 - s2:6d}")
        del A, Bm, C


def g6c_stamps(T=4096 * 300):
    """Slot timeline of one gemm6c workgroup (bf16x6 mode, library built with -DRLT_STAMPS): per wavefront and slot the cycles
    from slot start to its last MFMA issue, and the wait at the barrier behind it; even slots only multiply, odd slots also
    split the next register tile."""
    import ctypes
    fn = N.load().rlt_debug_g6c_stamps
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    N.set_precision("bf16x6")
    for name, ta, tb, M, Nn, K in [("ffn1 fwd NT", 0, 1, T, 2048, 256), ("ffn2 fwd NT", 0, 1, T, 256, 2048), ("ffn1 dW TN", 1, 0, 2048, 256, T)]:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        Bm = torch.randn((Nn, K) if tb else (K, Nn), device=dev)
        C = torch.empty(M, Nn, device=dev)
        ms = timeit(lambda: ops.gemm(ta, tb, M, Nn, K, A, A.shape[1], Bm, Bm.shape[1], C, Nn))
        buf = (ctypes.c_ulonglong * (8 * 40 * 3))()
        assert fn(buf, 8 * 40 * 3) == 0
        v = list(buf)
        print(f"gemm6c {name} {M}x{Nn}x{K}: {ms:.3f} ms; slots 2..33 of one workgroup: cycles [start->last MFMA issued | barrier wait], slot period")
        for w in range(8):
            row = []
            for sl in range(2, 34):
                t0, t1, t2 = v[(w * 40 + sl) * 3:(w * 40 + sl) * 3 + 3]
                nxt = v[(w * 40 + sl + 1) * 3]
                row.append(f"{t1 - t0:5d}|{t2 - t1:4d}|{nxt - t0:5d}")
            print(f"   w{w}: " + " ".join(row))
        ev = [v[(0 * 40 + sl + 1) * 3] - v[(0 * 40 + sl) * 3] for sl in range(2, 34, 2)]
        od = [v[(0 * 40 + sl + 1) * 3] - v[(0 * 40 + sl) * 3] for sl in range(3, 35, 2)]
        print(f"   wave 0 mean slot period: even {sum(ev) / len(ev):.0f}, odd {sum(od) / len(od):.0f} cycles (48 MFMAs per wavefront and slot = 3,072 matrix cycles per SIMD)")
        del A, Bm, C


def dkv1_stamps(B=4096, S=60, H=4, HD=64, which="dkv"):
    """Phase timeline of the one-wavefront-per-SIMD dK+dV kernel: needs a -DRLT_DKV1_STAMPS library (RLT_HIP_LIB)."""
    import ctypes
    E = H * HD
    T = S * B
    qkv = torch.randn(T, 3 * E, device=dev); out = torch.randn(T, E, device=dev); lse = torch.randn(S, H, B, device=dev).abs() + 20
    dout = torch.randn(T, E, device=dev); dqkv = torch.empty_like(qkv)
    ib = N.query("rlt_list_attention_fwd_workspace", S, B, H, HD, 0.0, N.PRECISION_DEFAULT)
    images = torch.empty(max(ib, 16) // 4, device=dev) if ib else None
    wb = N.query("rlt_list_attention_bwd_workspace", S, B, H, HD, 0.0, N.PRECISION_DEFAULT)
    ws = torch.empty(wb // 4 + 4, device=dev)
    call("rlt_list_attention_bwd_prepare", ptr(out), ptr(dout), ptr(lse), S, B, H, HD, 0.0, ptr(images), ptr(ws), wb, N.PRECISION_DEFAULT, stream())
    f = lambda: call("rlt_list_attention_bwd_" + which, ptr(qkv), ptr(dout), ptr(lse), ptr(images), ptr(ws), wb, S, B, H, HD, 0.0, 7, ptr(dqkv), N.PRECISION_DEFAULT, stream())
    ms = timeit(f)
    fn = N.load().rlt_debug_dkv1_stamps
    fn.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * (4 * 4 * 66))()
    assert fn(buf) == 0
    v = list(buf)
    nstep = 64 if which == "dkv" else 48
    print(f"{which}1 {ms:.3f} ms; cycles per step (six MFMAs = 192 matrix cycles) of tiles 9, 10 of wavefronts 0 and 3, then barrier wait and tile period")
    for w in (0, 3):
        for tl in (1, 2):
            st = v[(w * 4 + tl) * 66:(w * 4 + tl) * 66 + 66]
            nx = v[(w * 4 + tl + 1) * 66]
            d = [st[k + 1] - st[k] for k in range(nstep - 1)] + [st[64] - st[nstep - 1]]
            print(f"  w{w} t{8 + tl}: " + " ".join(f"{x:4d}" for x in d) + f" | {st[65] - st[64]:5d} | {nx - st[0]:6d}")


def a6n_stamps(which="dkv", B=8192, S=20, H=8, HD=16):
    """Slot timeline of the pipelined head-dim-16 backward kernels: needs a -DRLT_A6N_STAMPS library (RLT_HIP_LIB)."""
    import ctypes
    E = H * HD
    T = S * B
    qkv = torch.randn(T, 3 * E, device=dev); out = torch.randn(T, E, device=dev); lse = torch.randn(S, H, B, device=dev).abs() + 20
    dout = torch.randn(T, E, device=dev); dqkv = torch.empty_like(qkv)
    wb = N.query("rlt_list_attention_bwd_workspace", S, B, H, HD, 0.0, N.PRECISION_DEFAULT)
    ws = torch.empty(wb // 4 + 4, device=dev)
    call("rlt_list_attention_bwd_prepare", ptr(out), ptr(dout), ptr(lse), S, B, H, HD, 0.0, None, ptr(ws), wb, N.PRECISION_DEFAULT, stream())
    f = lambda: call("rlt_list_attention_bwd_" + which, ptr(qkv), ptr(dout), ptr(lse), None, ptr(ws), wb, S, B, H, HD, 0.0, 7, ptr(dqkv), N.PRECISION_DEFAULT, stream())
    ms = timeit(f)
    fn = N.load().rlt_debug_a6n_stamps
    fn.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * (4 * 4 * 18))()
    assert fn(buf) == 0
    v = list(buf)
    gs = 32 if which == "dkv" else 22
    print(f"{which} {ms:.3f} ms; cycles per slot ({gs} MFMAs = {16 * gs} matrix cycles) of tiles 9, 10 of wavefronts 0 and 3, then barrier wait and tile period")
    for w in (0, 3):
        for tl in (1, 2):
            st = v[(w * 4 + tl) * 18:(w * 4 + tl) * 18 + 18]
            nx = v[(w * 4 + tl + 1) * 18]
            d = [st[k + 1] - st[k] for k in range(16)]
            print(f"  w{w} t{8 + tl}: " + " ".join(f"{x:4d}" for x in d) + f" | {st[17] - st[16]:5d} | {nx - st[0]:6d}")


def a6h_stamps(which="fwd", B=4096, S=60, H=4, HD=64):
    """Slot timeline of the pipelined head-dim-64 kernels: needs a -DRLT_A6H_STAMPS library (RLT_HIP_LIB)."""
    import ctypes
    E = H * HD
    T = S * B
    qkv = torch.randn(T, 3 * E, device=dev); out = torch.randn(T, E, device=dev); lse = torch.randn(S, H, B, device=dev).abs() + 20
    dout = torch.randn(T, E, device=dev); dqkv = torch.empty_like(qkv)
    ib = N.query("rlt_list_attention_fwd_workspace", S, B, H, HD, 0.0, N.PRECISION_DEFAULT)
    images = torch.empty(max(ib, 16) // 4, device=dev) if ib else None
    wb = N.query("rlt_list_attention_bwd_workspace", S, B, H, HD, 0.0, N.PRECISION_DEFAULT)
    ws = torch.empty(wb // 4 + 4, device=dev)
    if which == "fwd":
        f = lambda: call("rlt_list_attention_fwd", ptr(qkv), S, B, H, HD, 0.0, 7, ptr(out), ptr(lse), ptr(images), ib, N.PRECISION_DEFAULT, stream())
    else:
        call("rlt_list_attention_bwd_prepare", ptr(out), ptr(dout), ptr(lse), S, B, H, HD, 0.0, ptr(images), ptr(ws), wb, N.PRECISION_DEFAULT, stream())
        f = lambda: call("rlt_list_attention_bwd_" + which, ptr(qkv), ptr(dout), ptr(lse), ptr(images), ptr(ws), wb, S, B, H, HD, 0.0, 7, ptr(dqkv), N.PRECISION_DEFAULT, stream())
    ms = timeit(f)
    fn = N.load().rlt_debug_a6h_stamps
    fn.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * (4 * 4 * 18))()
    assert fn(buf) == 0
    v = list(buf)
    gs, ns = {"fwd": (52, 8), "dq": (76, 4), "dkv": (104, 4)}[which]
    print(f"{which} {ms:.3f} ms; cycles per slot ({gs} MFMAs = {16 * gs} matrix cycles) of tiles 9, 10 of wavefronts 0 and 3, then the last slot up to the barrier, the barrier wait and the tile period")
    for w in (0, 3):
        for tl in (1, 2):
            st = v[(w * 4 + tl) * 18:(w * 4 + tl) * 18 + 18]
            nx = v[(w * 4 + tl + 1) * 18]
            d = [st[k + 1] - st[k] for k in range(ns - 1)] + [st[16] - st[ns - 1]]
            print(f"  w{w} t{8 + tl}: " + " ".join(f"{x:4d}" for x in d) + f" | {st[17] - st[16]:5d} | {nx - st[0]:6d}")


def a6h_stamps_dq():
    a6h_stamps(which="dq")


def a6h_stamps_dkv():
    a6h_stamps(which="dkv")


def a6n_stamps_dq():
    a6n_stamps(which="dq")


def dq1_stamps():
    dkv1_stamps(which="dq")


if __name__ == "__main__":
    which = sys.argv[1:] or ["attention", "gemms", "lstm"]
    print("env:", {k: v for k, v in os.environ.items() if k.startswith("RLT_")}, flush=True)
    for w in which:
        globals()[w]()
