cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r04_gemm6c_ab.log
echo "=== correctness: gemm + scale_ops sections in bf16x6, gemm6c non-persistent" > $O
RLT_GEMM6_PERSIST=0 timeout -k 10 500 python tools/gpu_probe.py --precision=bf16x6 gemm scale_ops 2>&1 | grep -v "^OK " >> $O
echo "=== correctness: same, persistent" >> $O
timeout -k 10 500 python tools/gpu_probe.py --precision=bf16x6 gemm scale_ops 2>&1 | grep -v "^OK " >> $O
for cfg in "RLT_GEMM6C=0" "RLT_GEMM6C=1 RLT_GEMM6_PERSIST=0" "RLT_GEMM6C=1 RLT_GEMM6_PERSIST=256"; do
  echo "=== timing: $cfg" >> $O
  env $cfg timeout -k 10 300 python tools/x6_probe.py --modes=bf16x6 >> $O 2>&1
done
cat $O | cut -c1-200
