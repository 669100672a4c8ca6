set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests_1.log 2>&1 || { tail -40 gpurun_out/r04_gpu_tests_1.log; exit 1; }
tail -45 gpurun_out/r04_gpu_tests_1.log
