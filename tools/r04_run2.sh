cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python tools/gpu_probe.py x6_adversarial precision_argument determinism bench_two_ranks trainer_dp_mt trainer_dp > gpurun_out/r04_new_sections.log 2>&1
echo "exit $?"
grep -n "^FAIL\|^--- \|ok, .* failed\|Traceback\|Error" gpurun_out/r04_new_sections.log | head -60
