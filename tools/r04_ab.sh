# GPU box: A/B of environment-switched kernel variants.  usage: r04_ab.sh TAG "ENV1" "ENV2" ... (each a space-separated env list)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=$1; shift
O=gpurun_out/r04_ab_$TAG.log
: > $O
for cfg in "$@"; do
  echo "=== correctness ($cfg): gemm scale_ops" >> $O
  env $cfg timeout -k 10 500 python tools/gpu_probe.py --precision=bf16x6 gemm scale_ops 2>&1 | grep "^FAIL\|ok, .* failed" >> $O
done
for ROUND in 1 2; do
for cfg in "$@"; do
  echo "=== timing round $ROUND: $cfg" >> $O
  env $cfg timeout -k 10 300 python tools/x6_probe.py --modes=bf16x6 2>&1 | grep -v amdgpu.ids >> $O
done
done
cat $O | cut -c1-140
