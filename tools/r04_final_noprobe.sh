# the measurement half of tools/final_run.sh (the GPU suite runs in its own call)
set -e
cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r04 > gpurun_out/r04_refresh.log 2>&1
for M in fp32 bf16x6 bf16x3; do
  PMC_SQ_ARGS="--precision $M" bash tools/pmc_sq_step.sh r04_$M > /dev/null 2>&1 && cp gpurun_out/r04_${M}_pmc_sq.txt gpurun_out/r04_pmc_sq_$M.txt
done
head -c 1500 gpurun_out/r04_bench_n1.json
