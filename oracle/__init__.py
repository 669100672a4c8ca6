"""CPU oracle for the ranked-list-truncation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`ranked-list-truncation_amd/`) may import from here.  The only legitimate
importers are `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py`, and there only as the checker / the timed CPU baseline.

What it is: a CPU restatement (torch-CPU fp32 for the models, numpy float64 for
the evaluation metrics) of the reference's hot path, SURVEY.md section 8(a):

    oracle/models.py    M1..M7  models/{AttnCut,Choopy,MtAttnCut,MtChoopy,MMOECut}.py
    oracle/explicit.py  M2,M3   the same LSTM / encoder-layer arithmetic spelled out
                                (gate order, batch-axis attention, post-norm) - documents
                                what the HIP kernels implement
    oracle/losses.py    L1..L8  utils/losses.py + utils/metrics.py:79-101
    oracle/metrics.py   E1..E3  utils/metrics.py:9-38, run.py:137-142
    oracle/weights.py           deterministic weight / input recipes (build-owned)

Pinning: the reference has no tests and pins no torch version, so the
oracle is pinned against outputs of the reference itself, imported in the build
container by `tools/make_golden.py` (torch 2.10.0 CPU), committed as fixtures under
`tests/golden/` and checked by `tests/test_oracle_golden.py`, plus the two known-answer
inputs the reference carries (utils/metrics.py:104-109).
"""
