#!/bin/bash
# GPU box: parity of the one-wavefront-per-SIMD dK+dV / dQ kernels and their launch times against the two-workgroup forms
R=$GRAFT_REPO_ROOT
TAG=${1:-dkv1}
timeout -k 10 300 python $R/tools/gpu_probe.py attention dropout --precision=bf16x6 > $R/gpurun_out/${TAG}_probe.log 2>&1
grep -n "FAIL\|ok, " $R/gpurun_out/${TAG}_probe.log | tail -12
RLT_PRECISION=bf16x6 timeout -k 10 200 python $R/tools/bench_kernels.py attention 2>&1 | grep attn_ > $R/gpurun_out/${TAG}_times.log
RLT_PRECISION=bf16x6 RLT_A6_DKV1=0 RLT_A6_DQ1=0 timeout -k 10 200 python $R/tools/bench_kernels.py attention 2>&1 | grep attn_bwd_d >> $R/gpurun_out/${TAG}_times.log
cat $R/gpurun_out/${TAG}_times.log
