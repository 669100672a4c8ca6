"""Parity of the HIP hot path (through the C ABI, on a real MI355X) against the CPU oracle, torch
fp64 references and the golden vectors the reference produced.  `pytest -m gpu`.

The checks live in tools/gpu_probe.py (one section per kernel family, every case reports
|error| vs tolerance); each test runs a section and requires every case to be within tolerance:

    gemm        rlt_gemm all transpose modes, ragged sizes, K=3, split-K, bias/ReLU/accumulate; rlt_colsum
    losses      every criterion x metric against the reference's golden losses and dL/dp (edge rows:
                no relevant doc, all relevant, single positive first/last), ragged S, MtCutLoss
    metrics     Metric.f1 / Metric.dcg known answers of the reference, argmax cut positions
    layernorm   residual+LayerNorm forward/backward vs fp64
    heads       softmax / sigmoid / identity heads forward/backward vs fp64
    attention   list-axis attention forward/backward vs fp64 (B not multiple of any tile, HD 16/32/64)
    lstm        2-layer BiLSTM forward/backward vs nn.LSTM
    embed_mmoe  Choopy embedding, MMOE gates and mixture forward/backward vs fp64
    dropout     the four dropout sites of the encoder layer against references built with the kernels'
                own keep-masks; keep-rate statistics; eval() ignores dropout; reproducible under manual_seed
    optimizer_and_trainer  FusedAdam vs torch.optim.Adam on the oracle over 3 real steps; run.py Trainer on a
                synthetic robust04-format set for 4 models, checkpoints load into reference-shaped modules
    models      all 18 golden model cases: outputs (1e-5), cut positions (identical), F1/DCG (1e-4),
                every criterion's loss (1e-4), per-parameter gradients (1e-3 of the gradient norm)
Tolerances are written next to each case in tools/gpu_probe.py; BASELINE.json asks for 1e-4.
"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
_SPEC = importlib.util.spec_from_file_location(
    "gpu_probe", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gpu_probe.py"))


@pytest.fixture(scope="module")
def probe():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    mod = importlib.util.module_from_spec(_SPEC)
    _SPEC.loader.exec_module(mod)
    from rlt_hip import native
    native.load()            # fails loudly if librlt_hip.so is missing: there is no fallback path
    return mod


MODE_DEPENDENT = ["gemm", "attention", "lstm", "dropout", "optimizer_and_trainer", "models", "bicut"]
MODE_FREE = ["losses", "metrics", "layernorm", "heads", "embed_mmoe"]


def _run(probe, name):
    import torch
    probe.RESULTS.clear()
    getattr(probe, name)()
    torch.cuda.synchronize()
    assert probe.RESULTS, "section produced no checks"
    bad = [(n, e, t) for (n, e, t, ok) in probe.RESULTS if not ok]
    assert not bad, f"{len(bad)} of {len(probe.RESULTS)} checks out of tolerance: {bad[:8]}"


@pytest.mark.parametrize("name", MODE_FREE)
def test_section(probe, name):
    _run(probe, name)


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
@pytest.mark.parametrize("name", MODE_DEPENDENT)
def test_section_by_precision(probe, name, precision):
    """Both MFMA precision modes of the library: the default split-bf16 mode and the exact-fp32 mode.
    Model-level tolerances are identical in both; op-level MFMA tolerances are 6x looser for bf16x3
    (tools/gpu_probe.py: mfma_tol)."""
    from rlt_hip import native
    native.set_precision(precision)
    try:
        _run(probe, name)
    finally:
        native.set_precision("bf16x3")


def test_full_size_properties(probe):
    """BASELINE-size invariants that need no oracle: at batch 4096 x len 300 every cut distribution
    sums to 1, is non-negative, and the loss/metrics are finite; the reward distribution q sums to 1."""
    import torch
    import models as hm
    from utils import losses as hl
    from utils.metrics import Metric
    from rlt_hip import ops, native as N
    import bench
    dev = torch.device("cuda")
    x, y = bench.synth_batch(4096, 300, 3, 7, dev)
    model = hm.AttnCut(dropout=0.0).to(dev)
    p = model(x)
    loss = hl.DivLoss(metric='f1', div_type='js', augmented=True)(p, y)
    loss.backward()
    sums = p.detach().squeeze(2).double().sum(1)
    assert float((sums - 1).abs().max()) < 1e-5
    assert float(p.min()) >= 0.0
    assert torch.isfinite(loss).item()
    for prm in model.parameters():
        assert torch.isfinite(prm.grad).all()
    k, f1, dcg = Metric.evaluate(p, y)
    assert int(k.min()) >= 1 and int(k.max()) <= 300 and 0.0 <= float(f1) <= 1.0
    r, q = ops.reward_matrix(y, N.METRIC_F1, tau=0.85, want_q=True)
    assert float((q.double().sum(1) - 1).abs().max()) < 1e-5
    assert float(r.min()) >= 0.0 and float(r.max()) <= 1.0
