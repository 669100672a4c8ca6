"""The host side of librlt_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5 "race detection /
sanitizers"): a HOST-ONLY build of every csrc/*.hip (`--offload-host-only`: argument checking, precision
scopes, workspace-layout arithmetic with -DRLT_BOUNDS_CHECK, split-K planning; no device code) is loaded in a child python
with clang's ASan runtime preloaded, and the no-GPU ABI tests (tests/test_abi.py) plus a walk over the path-level entry
points with real host buffers run against it.  Any sanitizer report aborts the child.  CPU box only: GPU ASan is not
available on this pool."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WALK = r"""
import ctypes, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/ranked-list-truncation_amd")
from rlt_hip import native as N
lib = N.load()
assert lib.rlt_abi_version() == 5
# workspace queries over a grid of shapes and every precision code: pure host arithmetic
for S, B, E, H, FF in [(300, 4096, 256, 4, 2048), (300, 8192, 128, 8, 2048), (40, 63, 256, 4, 2048), (1, 1, 64, 1, 64), (7, 33, 128, 8, 96)]:
    for prec in (-1, 0, 1, 2):
        for op in (1, 2, 3, 4, 5):
            for drop in (0, 1):
                n = N.query("rlt_workspace_bytes", op, S, B, E, H, FF, drop, prec)
                assert n % 256 == 0 and (n > 0 or op == 2), (op, S, B, E, H, FF, drop, prec, n)   # (op 2: split-K scratch, may be empty)
        assert N.query("rlt_list_attention_bwd_workspace", S, B, H, E // H, 0.0, prec) >= S * B * H * 4
    for ta, tb in ((0, 1), (0, 0), (1, 0), (1, 1)):
        N.query("rlt_gemm_workspace", ta, tb, S * B, FF, E)
        N.query("rlt_gemm_workspace", ta, tb, FF, E, S * B)
# undersized buffers: the path-level entry points must answer RLT_E_WORKSPACE (-3) before any layout pointer is formed
S, B, E, H, FF = 3, 5, 64, 2, 64
stash_b = N.query("rlt_workspace_bytes", N.OP_ENCODER_STASH, S, B, E, H, FF, 0, 0)
ws_b = N.query("rlt_workspace_bytes", N.OP_ENCODER_BWD_WS, S, B, E, H, FF, 0, 0)
buf = (ctypes.c_uint8 * (stash_b + ws_b + 4096))()
base = ctypes.addressof(buf)
w = N.EncoderPtrs(*[base] * 12)
x = ctypes.c_void_p(base)
assert lib.rlt_encoder_layer_fwd(x, ctypes.byref(w), S, B, E, H, FF, 1e-5, 0.0, None, x, x, stash_b - 1, x, 1 << 20, 0, None) == -3
assert lib.rlt_encoder_layer_bwd(x, ctypes.byref(w), S, B, E, H, FF, 1e-5, 0.0, None, x, x, stash_b, x, ctypes.byref(w), x, ws_b - 1, 0, None) == -3
lw = (N.LstmLayerPtrs * 2)()
for l in range(2):
    for f in ("w_ih", "w_hh", "b_ih", "b_hh"):
        getattr(lw[l], f)[0] = base
        getattr(lw[l], f)[1] = base
ls_b = N.query("rlt_workspace_bytes", N.OP_BILSTM_STASH, S, B, 3, 0, 0, 0, 0)
lw_b = N.query("rlt_workspace_bytes", N.OP_BILSTM_WS, S, B, 3, 0, 0, 0, 0)
assert lib.rlt_bilstm_fwd(x, 3, lw, S, B, x, x, ls_b - 1, x, lw_b, 0, None) == -3
assert lib.rlt_bilstm_fwd(x, 3, lw, S, B, x, x, ls_b, x, lw_b - 1, 0, None) == -3
assert lib.rlt_bilstm_bwd(x, 3, lw, x, x, S, B, x, ls_b - 1, None, lw, x, lw_b, 0, None) == -3
# ABI 4: the DCG coefficient table is caller memory - size query, undersized / misaligned table, and the argument checks of the
# training-step loss with real host buffers (every check answers before a launch is formed)
tb = N.query("rlt_dcg_table_bytes")
assert tb >= 2049 * 8 and tb % 16 == 0
assert lib.rlt_dcg_table_init(x, tb - 1, None) == -3
assert lib.rlt_dcg_table_init(ctypes.c_void_p(base + 4), tb, None) == -4
lm_b = N.query("rlt_loss_metrics_workspace", B)
assert lib.rlt_loss_metrics(x, x, None, B, S, 0, -1.0, 3, 0.85, -1.0, x, x, x, x, x, x, x, None, x, lm_b, None) == -1      # no table
assert lib.rlt_loss_metrics(x, x, None, B, S, 0, -1.0, 3, 0.85, -1.0, x, x, x, x, x, x, x, ctypes.c_void_p(base + 4), x, lm_b, None) == -4
assert lib.rlt_loss_metrics(x, x, None, B, S, 0, -1.0, 3, 0.85, -1.0, x, x, x, x, x, x, x, x, x, lm_b - 1, None) == -3
print("walk ok")
"""


@pytest.fixture(scope="module")
def sanitized():
    from rlt_hip import build
    rt = build.sanitizer_runtime()
    if rt is None:
        pytest.skip("clang's shared ASan runtime is not in this toolchain")
    lib = build.build_sanitized(verbose=False)
    env = dict(os.environ, LD_PRELOAD=rt, RLT_HIP_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1:detect_odr_violation=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    return env


def test_abi_tests_under_asan_ubsan(sanitized):
    # the ABI tests of the normal build, now against the sanitized library (RLT_HIP_LIB) in a child process
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_abi.py"), "-x", "-q", "-p", "no:cacheprovider",
                          "-k", "not fails_loudly and not mirror_reference"], env=sanitized, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in res.stderr and "runtime error:" not in res.stderr, res.stderr[-3000:]


def test_path_level_layout_walk_under_asan_ubsan(sanitized):
    res = subprocess.run([sys.executable, "-c", WALK, REPO], env=sanitized, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "walk ok" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in res.stderr and "runtime error:" not in res.stderr, res.stderr[-3000:]
