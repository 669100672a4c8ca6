#!/bin/bash
# GPU box: bf16x6 attention kernels at the AttnCut shape, ping-pong forward off / on (RLT_A6_PP), for each library variant; two rounds
cd $GRAFT_REPO_ROOT
for ROUND in 1 2; do
for V in "$@"; do
for PP in 0 1; do
  echo "== $V PP=$PP (round $ROUND)"
  RLT_A6_PP=$PP RLT_PRECISION=bf16x6 RLT_HIP_LIB=$GRAFT_REPO_ROOT/ranked-list-truncation_amd/csrc/variants/librlt_$V.so timeout -k 10 120 python3 tools/bench_kernels.py attention 2>&1 | grep "attn_fwd\|attn_bwd_d" || exit 1
done
done
done
