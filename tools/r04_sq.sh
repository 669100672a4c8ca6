cd $GRAFT_REPO_ROOT
for M in bf16x6 fp32 bf16x3; do
  PMC_SQ_ARGS="--precision $M" bash tools/pmc_sq_step.sh r04_$M > /dev/null 2>&1 && cp gpurun_out/r04_${M}_pmc_sq.txt gpurun_out/r04_pmc_sq_$M.txt
  echo "$M SQ pass done"
done
head -30 gpurun_out/r04_pmc_sq_bf16x6.txt
