set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r04_final_gpu_tests.log 2>&1 || { tail -20 gpurun_out/r04_final_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r04_final_gpu_tests.log
bash tools/refresh_profiles.sh r04 > gpurun_out/r04_refresh.log 2>&1
for M in fp32 bf16x6 bf16x3; do
  PMC_SQ_ARGS="--precision $M" bash tools/pmc_sq_step.sh r04_$M > /dev/null 2>&1 && cp gpurun_out/r04_${M}_pmc_sq.txt gpurun_out/r04_pmc_sq_$M.txt
done
bash tools/side_configs.sh r04 > /dev/null 2>&1
head -3 gpurun_out/r04_side_configs.txt
tail -c 300 gpurun_out/r04_bench_n1.json
