R=$GRAFT_REPO_ROOT
cd $R
python tools/gpu_probe.py losses metrics --precision=bf16x6 > gpurun_out/r6_loss.log 2>&1
echo "rc=$?"; grep -n "FAIL\|ok, \|EXCEPTION\|Traceback" gpurun_out/r6_loss.log | tail
for i in 1 2; do
echo "--- four lists per wavefront"; python tools/bench_kernels.py loss 2>&1 | grep "fused"
echo "--- two lists per wavefront (RLT_LOSS_QUARTERS=0)"; RLT_LOSS_QUARTERS=0 python tools/bench_kernels.py loss 2>&1 | grep "fused"
done
