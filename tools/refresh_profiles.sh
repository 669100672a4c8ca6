#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh TAG'): the default bench line (exact-fp32 headline + bf16x3
# fast_mode block), rocprofv3 kernel stats and per-call traces of the AttnCut step in BOTH precision modes and of the Choopy
# step, and the two PMC traffic passes per mode.  Outputs under gpurun_out/TAG_*; tools/collect_profiles.py TAG then
# writes the summaries that are committed under profiles/.
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
tail -c 600 $O/${TAG}_bench_n1.json; echo
export RLT_BENCH_SMALL=0      # (the profiled runs: the headline step only, not the small-batch block of the bench line)
for MODE in fp32 bf16x3 bf16x6; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_attncut_$MODE -o p -- python3 $R/bench.py --precision $MODE --steps 6 --warmup 2 --no-cpu-baseline --other-steps 0 > $O/${TAG}_prof_attncut_$MODE.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch_$MODE -o p -- python3 $R/bench.py --precision $MODE --steps 2 --warmup 1 --no-cpu-baseline --other-steps 0 > $O/${TAG}_pmc_fetch_$MODE.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_write_$MODE -o p -- python3 $R/bench.py --precision $MODE --steps 2 --warmup 1 --no-cpu-baseline --other-steps 0 > $O/${TAG}_pmc_write_$MODE.log 2>&1
  echo "$MODE profiled"
done
if [ "$2" != "nochoopy" ]; then
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_choopy -o p -- python3 $R/bench.py --precision bf16x3 --model choopy --batch 8192 --steps 4 --warmup 2 --no-cpu-baseline --other-steps 0 > $O/${TAG}_prof_choopy.log 2>&1
  tail -c 400 $O/${TAG}_prof_choopy.log; echo
fi
ls $O/${TAG}_prof_attncut_fp32 $O/${TAG}_pmc_fetch_fp32
