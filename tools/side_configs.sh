#!/bin/bash
# Runs on the GPU box: bench.py --no-cpu-baseline for the side configurations of SURVEY 8(d) (C3, C4, C5, conf dropout);
# one JSON line per run into gpurun_out/TAG_side_configs.log
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_side_configs.log
echo "# bench.py --no-cpu-baseline --fp32-steps 0 on one MI355X, side configurations of SURVEY 8(d); one JSON line per run" > $O
run() { timeout -k 10 400 python3 $R/bench.py --no-cpu-baseline --fp32-steps 0 "$@" >> $O 2>> $O.err || echo "FAILED: $*" >> $O; }
run --batch 32 --steps 40 --warmup 5          # (first: the first process on a fresh box pages the image in)
run --model mmoecut --num-tasks 2.1 --batch 2048 --steps 3 --warmup 1
run --model mmoecut --num-tasks 2.2 --batch 2048 --steps 3 --warmup 1
run --model mmoecut --num-tasks 2.1 --batch 2048 --steps 3 --warmup 1 --dropout 0.2
run --model mtattncut --num-tasks 3 --buckets 100,200,300 --steps 6 --warmup 2
run --model mtattncut --num-tasks 3 --buckets 100,200,300 --reward dcg --steps 6 --warmup 2
run --model mtattncut --num-tasks 3 --steps 5 --warmup 2
run --dropout 0.4 --steps 5 --warmup 2
run --model choopy --batch 8192 --steps 3 --warmup 1
run --model choopy --batch 8192 --steps 3 --warmup 1 --dropout 0.2
run --model choopy --batch 32 --steps 40 --warmup 5
run --batch 63 --steps 40 --warmup 5
tail -c 300 $O
