#!/bin/bash
# GPU box: time the head-dim-16 six-product attention kernels (Choopy shape, 20 positions) for each library variant named on the
# command line ("base" = the product library), two rounds interleaved
cd $GRAFT_REPO_ROOT
for R in 1 2; do
for V in "$@"; do
  echo "== $V (round $R)"
  if [ "$V" = base ]; then LIB=$GRAFT_REPO_ROOT/ranked-list-truncation_amd/csrc/librlt_hip.so; else LIB=$GRAFT_REPO_ROOT/ranked-list-truncation_amd/csrc/variants/librlt_$V.so; fi
  RLT_HIP_LIB=$LIB timeout -k 10 120 python3 tools/bench_kernels.py attention16 2>&1 | grep attn_ || exit 1
done
done
