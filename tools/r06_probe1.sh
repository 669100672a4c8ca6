R=$GRAFT_REPO_ROOT
cd $R
python tools/gpu_probe.py attention scale_attention path_level precision_argument x6_fallbacks x6_adversarial full_size_oracle_choopy rccl_two_ranks --precision=bf16x6 > gpurun_out/r6_probe1.log 2>&1
echo "rc=$?"
grep -n "FAIL\|ok, \|EXCEPTION\|Traceback\|^---" gpurun_out/r6_probe1.log | tail -40
grep -n "rccl_two_ranks\|full_size_oracle_choopy" gpurun_out/r6_probe1.log | tail -20
