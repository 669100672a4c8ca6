#!/bin/bash
# GPU box: time the bf16x6 list-attention kernels (AttnCut shape and Choopy shape) for each library variant, two interleaved rounds
cd $GRAFT_REPO_ROOT
for ROUND in 1 2; do
for V in "$@"; do
  echo "== $V (round $ROUND)"
  RLT_PRECISION=bf16x6 RLT_HIP_LIB=$GRAFT_REPO_ROOT/ranked-list-truncation_amd/csrc/variants/librlt_$V.so timeout -k 10 120 python3 tools/bench_kernels.py attention attention16 2>&1 | grep "attn_fwd\|attn_bwd_d" || exit 1
done
done
