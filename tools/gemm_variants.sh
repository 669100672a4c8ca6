#!/bin/bash
# GPU box: time the fp32 GEMM shapes of the AttnCut step for each library variant named on the command line, two interleaved rounds
cd $GRAFT_REPO_ROOT
for ROUND in 1 2; do
for V in "$@"; do
  echo "== $V (round $ROUND)"
  RLT_PRECISION=${GEMM_PRECISION:-fp32} RLT_HIP_LIB=$GRAFT_REPO_ROOT/ranked-list-truncation_amd/csrc/variants/librlt_$V.so timeout -k 10 120 python3 tools/bench_kernels.py gemms 2>&1 | grep "^gemm" || exit 1
done
done
