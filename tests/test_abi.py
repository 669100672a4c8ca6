"""The C-ABI library builds for gfx950, loads without a GPU, and exports every symbol that
include/rlt_hip.h declares (no compute calls here).  The product surface mirrors the reference's."""
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from rlt_hip import build, native
    build.build(verbose=False)          # hipcc cross-compiles without a GPU
    native.load()
    return native


def test_header_symbols_are_exported_and_bound(native):
    header = open(os.path.join(REPO, "include", "rlt_hip.h")).read()
    declared = set(re.findall(r"\b(rlt_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(native.EXPORTS), (declared ^ set(native.EXPORTS))
    lib = native.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rlt_abi_version() == 5
    assert b"workspace" in lib.rlt_error_string(-3)


def test_workspace_queries_need_no_gpu(native):
    assert native.query("rlt_gemm_workspace", 1, 0, 2048, 256, 1228800) > 0     # split-K slabs for dW
    assert native.query("rlt_gemm_workspace", 0, 1, 1228800, 2048, 256) == 0
    assert native.query("rlt_list_attention_bwd_workspace", 300, 4096, 4, 64, 0.0, native.PRECISION_DEFAULT) >= 300 * 4096 * 4 * 4
    assert native.query("rlt_colsum_workspace", 1000, 64) > 0


def test_argument_errors_are_reported_not_crashed(native):
    lib = native.load()
    assert lib.rlt_gemm(0, 1, 0, 8, 8, None, 8, None, 8, None, 8, None, None, 0, None, 0, -1, None) == -1
    assert lib.rlt_list_attention_fwd(None, 1, 1, 1, 64, 0.0, 0, None, None, None, 0, -1, None) == -1
    assert lib.rlt_heads_fwd(None, None, None, None, 1, 1, 1, 64, None, None) == -1
    # entry points added for the narrow (input_size <= 3) LSTM layer and the 1-bit ReLU mask
    assert lib.rlt_bilstm_rec_fwd_x(None, 3, None, None, None, None, None, None, None, None, 1, 1, None, None, None, -1, None) == -1
    assert lib.rlt_narrow_dw(None, 4, None, 3, 3, 1, 4, None, None, None, 0, None) == -1
    assert lib.rlt_gemm_bits(0, 1, 8, 32, 8, None, 8, None, 8, None, 32, None, 0, 0.0, 0, None, None, 1.0, -1, None) == -1
    assert lib.rlt_pair_softmax_fwd(None, 1, 1, 0.0, 0, None, None) == -1
    assert lib.rlt_wass_loss_fwd(None, None, 2, 2, 1e-3, 100, 0.1, None, None, 0, None) == -1
    assert native.query("rlt_wass_loss_workspace", 63, 100) > 2 * 63 * 63 * 4
    assert lib.rlt_bicut_loss(None, None, 1, 1, 1, 0.65, 0.1, None, None, None, None) == -1
    assert native.query("rlt_narrow_dw_workspace", 1228800, 1024) > 0
    # the fused loss + metrics pass and the penalty forms
    assert lib.rlt_loss_metrics(None, None, None, 1, 1, 0, -1.0, 3, 0.85, -1.0, None, None, None, None, None, None, None, None, None, 0, None) == -1
    # the DCG coefficient table is caller memory (ABI 4): size query, argument checks of the fill
    assert native.query("rlt_dcg_table_bytes") == (2049 * 8 + 15) // 16 * 16
    assert lib.rlt_dcg_table_init(None, 1 << 20, None) == -1
    assert native.query("rlt_loss_metrics_workspace", 4096) == 1024 * 4 * 3 * 8
    assert lib.rlt_cut_metrics_ex(None, None, None, 1, 1, -1.0, None, None, None, None, None) == -1


def test_path_level_entry_points(native):
    """rlt_encoder_layer_fwd/bwd, rlt_bilstm_fwd/bwd, rlt_workspace_bytes: sizes are consistent with the
    documented stash layout and bad arguments are reported (no GPU needed)."""
    N = native
    lib = N.load()
    S, B, E, H, FF = 300, 4096, 256, 4, 2048
    T = S * B
    D = N.PRECISION_DEFAULT
    stash = N.query("rlt_workspace_bytes", N.OP_ENCODER_STASH, S, B, E, H, FF, 0, D)
    rup = lambda n: (n + 255) // 256 * 256
    floats = rup(T * 3 * E * 4) + 4 * rup(T * E * 4) + rup(S * H * B * 4) + 2 * rup(T * 2 * 4) + rup(T * FF * 4)
    bits = rup(N.query("rlt_gemm_bits_words", T, FF) * 4)
    # attention tile images live in the stash only where the backward reads them (ABI 5: rlt_list_attention_images_retained);
    # the images of the pipelined bf16x6 forward kernels are scratch of the forward call, sized by the FWD_WS query
    img_b = N.query("rlt_list_attention_fwd_workspace", S, B, H, E // H, 0.0, D)
    kept = lib.rlt_list_attention_images_retained(S, B, H, E // H, D)
    assert stash == floats + bits + (rup(img_b) if kept else 0)
    fwd_ws = N.query("rlt_workspace_bytes", N.OP_ENCODER_FWD_WS, S, B, E, H, FF, 0, D)
    fwd_ws_train = N.query("rlt_workspace_bytes", N.OP_ENCODER_FWD_WS, S, B, E, H, FF, 1, D)
    assert fwd_ws - fwd_ws_train == (0 if kept else rup(img_b) - rup(N.query("rlt_list_attention_fwd_workspace", S, B, H, E // H, 0.1, D)))
    if N.get_precision() == "bf16x6" and not any(os.environ.get(v) == "0" for v in ("RLT_A6H", "RLT_ATTN6")) and not os.environ.get("RLT_ATTN_MODE"):
        assert not kept and img_b > 2 * S * H * (B // 64) * 24576 and N.query("rlt_list_attention_fwd_workspace", S, B, H, E // H, 0.1, D) == img_b
        img16 = N.query("rlt_list_attention_fwd_workspace", S, 8192, 8, 16, 0.0, D)          # head dim 16: the pipelined forward has no train-mode form
        assert img16 > 0 and N.query("rlt_list_attention_fwd_workspace", S, 8192, 8, 16, 0.1, D) == 0
    # the backward parts check their workspace themselves (ABI 5): a delta-only buffer is refused, not overrun, where images are staged
    assert lib.rlt_list_attention_bwd_dkv(None, None, None, None, None, 0, 1, 1, 1, 64, 0.0, 0, None, D, None) == -1
    ws0 = N.query("rlt_workspace_bytes", N.OP_ENCODER_BWD_WS, S, B, E, H, FF, 0, D)
    ws1 = N.query("rlt_workspace_bytes", N.OP_ENCODER_BWD_WS, S, B, E, H, FF, 1, D)
    assert ws1 - ws0 >= T * E * 4                       # train-mode dropout keeps the branch gradients apart
    assert ws0 >= T * E * 4 + T * FF * 4                # dz2 + dhid
    assert N.query("rlt_workspace_bytes", N.OP_BILSTM_STASH, S, B, 3, 0, 0, 0, D) == 2 * (rup(T * 1024 * 4) + rup(T * 256 * 4)) + rup(T * 256 * 4)
    assert N.query("rlt_workspace_bytes", N.OP_BILSTM_WS, S, B, 3, 0, 0, 0, D) > T * 256 * 4
    assert N.query("rlt_workspace_bytes", 99, S, B, E, H, FF, 0, D) == 0
    assert N.query("rlt_workspace_bytes", N.OP_ENCODER_STASH, S, B, 250, 4, FF, 0, D) == 0          # E % H != 0
    assert lib.rlt_encoder_layer_fwd(None, None, 1, 1, 64, 1, 64, 1e-5, 0.0, None, None, None, 0, None, 0, D, None) == -1
    assert lib.rlt_encoder_layer_bwd(None, None, 1, 1, 64, 1, 64, 1e-5, 0.0, None, None, None, 0, None, None, None, 0, D, None) == -1
    assert lib.rlt_bilstm_fwd(None, 3, None, 1, 1, None, None, 0, None, 0, D, None) == -1
    assert lib.rlt_bilstm_bwd(None, 3, None, None, None, 1, 1, None, 0, None, None, None, 0, D, None) == -1
    assert lib.rlt_bilstm_generic_fwd(None, 3, 64, None, 1, 1, None, None, 0, None, 0, D, None) == -1
    assert lib.rlt_bilstm_generic_bwd(None, 3, 64, None, None, None, 1, 1, None, 0, None, None, None, 0, D, None) == -1
    assert N.query("rlt_bilstm_generic_bytes", 1, 300, 8, 3, 64) == 2 * (rup(2400 * 512 * 4) + rup(2400 * 128 * 4)) + rup(2400 * 128 * 4)


def test_precision_is_a_call_argument_and_the_default_is_reference_faithful(native):
    """ABI v3: the mode is an argument of every entry point that depends on it; the process
    default is only what RLT_PRECISION_DEFAULT resolves to, it is bf16x6 (fp32-faithful) unless the environment or
    rlt_set_precision says otherwise, and a bad code is an argument error.  Workspace layouts follow the ARGUMENT: the
    split-bf16 tile records exist in bf16x3 mode only, whatever the default is."""
    N = native
    lib = N.load()
    keep = lib.rlt_get_precision()
    try:
        if "RLT_PRECISION" not in os.environ:
            assert N.get_precision() == "bf16x6"
        S, B, H, HD = 300, 4096, 4, 64
        for default in ("fp32", "bf16x3", "bf16x6"):
            N.set_precision(default)
            assert N.get_precision() == default
            img3 = N.query("rlt_list_attention_fwd_workspace", S, B, H, HD, 0.0, N.PRECISION_BF16X3)
            assert img3 > 0 and lib.rlt_list_attention_images_retained(S, B, H, HD, N.PRECISION_BF16X3) == 1     # Q / K / V tile records
            assert N.query("rlt_list_attention_fwd_workspace", S, B, H, HD, 0.0, N.PRECISION_FP32) == 0
            img6 = N.query("rlt_list_attention_fwd_workspace", S, B, H, HD, 0.0, N.PRECISION_BF16X6)       # call-local K / V images of the pipelined forward
            assert lib.rlt_list_attention_images_retained(S, B, H, HD, N.PRECISION_BF16X6) == 0
            assert N.query("rlt_list_attention_fwd_workspace", S, B, H, HD, 0.0, N.PRECISION_DEFAULT) == {"bf16x3": img3, "bf16x6": img6, "fp32": 0}[default]
            st3 = N.query("rlt_workspace_bytes", N.OP_ENCODER_STASH, S, B, 256, H, 2048, 0, N.PRECISION_BF16X3)
            st6 = N.query("rlt_workspace_bytes", N.OP_ENCODER_STASH, S, B, 256, H, 2048, 0, N.PRECISION_BF16X6)
            assert st3 - st6 == (img3 + 255) // 256 * 256
        # a code outside {-1, 0, 1, 2}: argument error / 0 bytes, before anything else is looked at
        assert lib.rlt_gemm(0, 1, 8, 8, 8, None, 8, None, 8, None, 8, None, None, 0, None, 0, 7, None) == -1
        assert lib.rlt_encoder_layer_fwd(None, None, 1, 1, 64, 1, 64, 1e-5, 0.0, None, None, None, 0, None, 0, 3, None) == -1
        assert N.query("rlt_workspace_bytes", N.OP_ENCODER_STASH, S, B, 256, H, 2048, 0, 5) == 0
        assert lib.rlt_set_precision(-1) == -1 and lib.rlt_set_precision(3) == -1
        with pytest.raises(ValueError):
            N.precision_code("fp64")
    finally:
        lib.rlt_set_precision(keep)


def test_precision_scope_of_the_python_wrappers(native):
    """ops.precision(...) scopes nest and restore; outside any scope the process default is what a forward would use."""
    from rlt_hip import ops
    N = native
    keep = N.load().rlt_get_precision()
    try:
        N.set_precision("bf16x6")
        assert ops.current_precision() == N.PRECISION_BF16X6
        with ops.precision("fp32"):
            assert ops.current_precision() == N.PRECISION_FP32
            with ops.precision("bf16x3"):
                assert ops.current_precision() == N.PRECISION_BF16X3
            assert ops.current_precision() == N.PRECISION_FP32
            N.set_precision("bf16x3")                      # the default does not leak into a scope
            assert ops.current_precision() == N.PRECISION_FP32
        assert ops.current_precision() == N.PRECISION_BF16X3
    finally:
        N.load().rlt_set_precision(keep)


def test_models_mirror_reference_state_dict():
    import models as hm
    from oracle import models as om
    cfgs = [("AttnCut", {}), ("Choopy", {}), ("MtAttnCut", {}), ("MtChoopy", {"num_tasks": 2.1}),
            ("MMOECut", {}), ("MMOECut", {"num_experts": 4, "num_tasks": 2.2}),
            ("MOECut", {}), ("MOECut", {"num_experts": 4, "num_tasks": 2.1, "seq_len": 40}), ("PLECut", {}), ("PLECut", {"seq_len": 40}), ("BiCut", {"input_size": 3}), ("BiCut", {"input_size": 5, "fc_dimensions": 128})]
    for name, kw in cfgs:
        a, b = getattr(hm, name)(**kw), getattr(om, name)(**kw)
        ka = [(k, tuple(v.shape)) for k, v in a.state_dict().items()]
        kb = [(k, tuple(v.shape)) for k, v in b.state_dict().items()]
        assert ka == kb, name
        a.load_state_dict(b.state_dict())       # checkpoints are interchangeable


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import models as hm
    from utils import losses as hl
    with pytest.raises(RuntimeError, match="GPU"):
        hm.AttnCut(dropout=0.0)(torch.zeros(2, 300, 3))
    with pytest.raises(RuntimeError, match="GPU"):
        hl.DivLoss()(torch.full((2, 300, 1), 1 / 300), torch.zeros(2, 300))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "ranked-list-truncation_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src, os.path.join(root, f)
