#!/bin/bash
# GPU box: parity (attention section) and launch times of attention kernel variants (csrc/variants/librlt_dkv1_*.so), two rounds
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-dkv1_variants}.log
: > $OUT
for lib in "" $R/ranked-list-truncation_amd/csrc/variants/librlt_dkv1_*.so; do
  RLT_HIP_LIB=$lib timeout -k 10 300 python $R/tools/gpu_probe.py attention --precision=bf16x6 2>&1 | grep "ok, " | sed "s|^|$(basename ${lib:-product}) |" >> $OUT
done
for round in 1 2; do
  for lib in "" $R/ranked-list-truncation_amd/csrc/variants/librlt_dkv1_*.so; do
    name=$(basename "${lib:-product}")
    ms=$(RLT_HIP_LIB=$lib timeout -k 10 120 python $R/tools/bench_kernels.py attention 2>&1 | grep "attn_bwd_d[kq]" | awk '{printf "%s ", $5}')
    echo "$round $name dkv/dq $ms" >> $OUT
  done
done
cat $OUT
