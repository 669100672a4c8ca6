R=$GRAFT_REPO_ROOT
cd $R
python tools/gpu_probe.py dropout scale_dropout attention --precision=bf16x6 > gpurun_out/r6_probe2.log 2>&1
echo "rc=$?"
grep -n "FAIL\|ok, \|EXCEPTION\|Traceback\|^---" gpurun_out/r6_probe2.log | tail -20
echo "--- dropout 0.4, new forward"; python tools/bench_kernels.py attention_drop 2>&1 | grep -v "amdgpu.ids\|^env" | head -5
echo "--- dropout 0.4, RLT_A6H=0"; RLT_A6H=0 python tools/bench_kernels.py attention_drop 2>&1 | grep -v "amdgpu.ids\|^env" | head -5
