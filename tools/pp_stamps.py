#!/usr/bin/env python3
"""Diagnostic (GPU box): segment timeline of the ping-pong bf16x6 forward kernel - needs a library built with -DRLT_PP_STAMPS
(tools/build_variant.py ppst attention6.hip -DRLT_PP_STAMPS -fno-slp-vectorize) selected with RLT_HIP_LIB, and RLT_A6_PP=1."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "ranked-list-truncation_amd"))
import torch
from rlt_hip import native as N
from rlt_hip.native import call, ptr, stream
N.set_precision("bf16x6")
dev = torch.device("cuda")
B, S, H, HD = 4096, 8, 4, 64
E = H * HD
qkv = torch.randn(S * B, 3 * E, device=dev); out = torch.empty(S * B, E, device=dev); lse = torch.empty(S, H, B, device=dev)
ib = N.query("rlt_list_attention_fwd_workspace", S, B, H, HD, 0.0, N.PRECISION_DEFAULT)
images = torch.empty(max(ib, 16) // 4, device=dev) if ib else None
print("image workspace bytes:", ib)
for _ in range(3):
    call("rlt_list_attention_fwd", ptr(qkv), S, B, H, HD, 0.0, 7, ptr(out), ptr(lse), ptr(images), ib, N.PRECISION_DEFAULT, stream())
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["RLT_HIP_LIB"])
buf = (ctypes.c_ulonglong * 128)()
fn = getattr(lib, "rlt_debug_pp_stamps"); fn.restype = ctypes.c_int
assert fn(buf) == 0
v = list(buf)
names = ["X", "bar", "Y", "bar", "Z", "bar", "W", "bar"]
t0 = min(x for x in v if x)
for g in range(2):
    print("group", "AB"[g])
    for t in range(8):
        st = v[(g * 8 + t) * 8:(g * 8 + t) * 8 + 8]
        nxt = v[(g * 8 + t + 1) * 8] if t < 7 else None
        durs = [st[i + 1] - st[i] for i in range(7)] + ([nxt - st[7]] if nxt else [])
        print(f"  tile {8 + t}: start {st[0] - t0:7d}  " + "  ".join(f"{n} {d}" for n, d in zip(names, durs)))
