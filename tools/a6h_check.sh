#!/bin/bash
# GPU box: parity of the pipelined head-dim-64 kernels (attention6h.hip) against fp64 and their launch times at AttnCut's shape
# (4096 lists, 4 heads x 64, 60 of the 300 positions) next to attention6.hip's kernels (RLT_A6H=0).  bash tools/a6h_check.sh TAG [sections]
TAG=${1:-a6h}
SECS=${2:-attention scale_attention}
R=$GRAFT_REPO_ROOT
python $R/tools/gpu_probe.py $SECS --precision=bf16x6 > $R/gpurun_out/${TAG}_probe.log 2>&1
grep -n "FAIL\|ok, \|Error\|error" $R/gpurun_out/${TAG}_probe.log | tail -12
grep "HD64" $R/gpurun_out/${TAG}_probe.log | grep "B4096\|B512\|B576\|B832" 
echo "--- attention6h (16x16x32, pipelined)"; python $R/tools/bench_kernels.py attention 2>&1 | tee $R/gpurun_out/${TAG}_bench_new.log
echo "--- attention6 (32x32x16, RLT_A6H=0)"; RLT_A6H=0 python $R/tools/bench_kernels.py attention 2>&1 | tee $R/gpurun_out/${TAG}_bench_old.log
echo "--- attention6h again"; python $R/tools/bench_kernels.py attention 2>&1 | tee -a $R/gpurun_out/${TAG}_bench_new.log
