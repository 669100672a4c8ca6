#!/bin/bash
# GPU box: parity of the head-dim-16 six-product kernels (attention6n.hip) and their launch times at Choopy's shape next to the
# 32x32x16 kernels they replace (RLT_A6N=0) and the exact-fp32 kernels.  bash tools/a6n_check.sh TAG [sections]
TAG=${1:-a6n}
SECS=${2:-attention scale_ops}
R=$GRAFT_REPO_ROOT
python $R/tools/gpu_probe.py $SECS --precision=bf16x6 > $R/gpurun_out/${TAG}_probe.log 2>&1
grep -n "FAIL\|ok, \|Error\|error" $R/gpurun_out/${TAG}_probe.log | tail -12
echo "--- attention6n (16x16x32)"; python $R/tools/bench_kernels.py attention16 2>&1 | tee $R/gpurun_out/${TAG}_bench_new.log
echo "--- attention6 (32x32x16, RLT_A6N=0)"; RLT_A6N=0 python $R/tools/bench_kernels.py attention16 2>&1 | tee $R/gpurun_out/${TAG}_bench_old.log
echo "--- exact fp32 (attention16.hip)"; RLT_PRECISION=fp32 python $R/tools/bench_kernels.py attention16 2>&1 | tee $R/gpurun_out/${TAG}_bench_f32.log
