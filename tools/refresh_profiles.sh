#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh TAG'): the default bench line, rocprofv3 kernel stats and
# per-call tables of the AttnCut and Choopy steps, and the two PMC traffic passes.  Outputs under gpurun_out/TAG_*.
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
tail -c 600 $O/${TAG}_bench_n1.json; echo
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_attncut -o p -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --fp32-steps 0 > $O/${TAG}_prof_attncut.log 2>&1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_choopy -o p -- python3 $R/bench.py --model choopy --batch 8192 --steps 4 --warmup 2 --no-cpu-baseline --fp32-steps 0 > $O/${TAG}_prof_choopy.log 2>&1
tail -c 400 $O/${TAG}_prof_choopy.log; echo
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --fp32-steps 0 > $O/${TAG}_pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_write -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --fp32-steps 0 > $O/${TAG}_pmc_write.log 2>&1
ls $O/${TAG}_prof_attncut $O/${TAG}_pmc_fetch
