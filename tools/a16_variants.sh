#!/bin/bash
# GPU box: time the head-dim-16 fp32 attention kernels (Choopy shape, 20 positions) for each library variant named on the command line
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  echo "== $V"
  RLT_PRECISION=fp32 RLT_HIP_LIB=$GRAFT_REPO_ROOT/ranked-list-truncation_amd/csrc/variants/librlt_$V.so timeout -k 10 120 python3 tools/bench_kernels.py attention16 2>&1 | grep attn_ || exit 1
done
