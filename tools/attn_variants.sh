#!/bin/bash
# GPU box: time the fp32 list-attention kernels at the AttnCut shape (60 positions) for each library variant named on the command line,
# twice each, interleaved (clock / device drift shows as the spread between the two rounds)
cd $GRAFT_REPO_ROOT
FN=${ATTN_FN:-attention}
for ROUND in 1 2; do
for V in "$@"; do
  echo "== $V (round $ROUND)"
  RLT_PRECISION=fp32 RLT_HIP_LIB=$GRAFT_REPO_ROOT/ranked-list-truncation_amd/csrc/variants/librlt_$V.so timeout -k 10 120 python3 tools/bench_kernels.py $FN 2>&1 | grep attn_ || exit 1
done
done
