# GPU box: timing-only A/B of environment-switched variants.  usage: r04_ab_fast.sh TAG "ENV1" "ENV2" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=$1; shift
O=gpurun_out/r04_ab_$TAG.log
: > $O
for ROUND in 1 2; do
for cfg in "$@"; do
  echo "=== timing round $ROUND: $cfg" >> $O
  env $cfg timeout -k 10 300 python tools/x6_probe.py --modes=bf16x6 2>&1 | grep "sum of" >> $O
done
done
cat $O
