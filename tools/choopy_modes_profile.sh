#!/bin/bash
# Runs on the GPU box: kernel traces of the Choopy batch-8192 step in the fp32 and bf16x6 modes ->
# gpurun_out/TAG_choopy_{fp32,bf16x6}_per_call.txt
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for MODE in ${2:-fp32 bf16x6}; do
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_choopy_$MODE -o p -- python3 $R/bench.py --precision $MODE --model choopy --batch 8192 --steps 2 --warmup 1 --no-cpu-baseline --other-steps 0 > $O/${TAG}_prof_choopy_$MODE.log 2>&1
  python3 $R/tools/trace_calls.py $(find $O/${TAG}_prof_choopy_$MODE -name p_kernel_trace.csv | head -1) > $O/${TAG}_choopy_${MODE}_per_call.txt
  head -12 $O/${TAG}_choopy_${MODE}_per_call.txt
done
