#!/bin/bash
# GPU box: A/B of the default-mode training step on ONE box - `bash tools/ab_step.sh VAR` times bench.py (headline config, default
# mode only) with VAR=0 and VAR=1 (unset), alternating twice; boxes differ by a few percent, so only same-box pairs compare
cd $GRAFT_REPO_ROOT
VAR=${1:-RLT_GEMM6S}
for R in 1 2; do
  for V in 0 1; do
    env $VAR=$V RLT_BENCH_SMALL=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --other-steps 0 --steps 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$V', d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['other_kernels_ms'])" || exit 1
  done
done
