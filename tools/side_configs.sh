#!/bin/bash
# Runs on the GPU box: bench.py --no-cpu-baseline for the side configurations of SURVEY 8(d) (C3, C4, C5, conf dropout, the
# reference's own batch sizes) in each precision mode; raw JSON lines into gpurun_out/TAG_side_configs.log, one summary line per
# run (mode | arguments | ms/step | lists/s) into gpurun_out/TAG_side_configs.txt
TAG=${1:-r03}
MODES=${2:-"fp32 bf16x6 bf16x3"}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_side_configs.log
T=$R/gpurun_out/${TAG}_side_configs.txt
echo "# bench.py --no-cpu-baseline --other-steps 0 --precision MODE on one MI355X, side configurations of SURVEY 8(d); one JSON line per run" > $O
echo "# mode | bench.py arguments | ms/step | lists/s   (one MI355X; headline configuration first)" > $T
run() {
  local mode=$1; shift
  local line
  line=$(timeout -k 10 500 python3 $R/bench.py --no-cpu-baseline --other-steps 0 --precision $mode "$@" 2>> $O.err) || { echo "FAILED: $mode $*" >> $T; return; }
  echo "$line" >> $O
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$mode | $* | %.2f | %.0f' % (d['ms_per_step'], d['value']))" >> $T
}
python3 $R/bench.py --no-cpu-baseline --other-steps 0 --batch 32 --steps 5 --warmup 2 > /dev/null 2>&1    # (pages the image in)
for M in $MODES; do
  run $M --steps 5 --warmup 2
  run $M --dropout 0.4 --steps 5 --warmup 2
  run $M --model mmoecut --num-tasks 2.1 --batch 2048 --steps 3 --warmup 1
  run $M --model mmoecut --num-tasks 2.2 --batch 2048 --steps 3 --warmup 1
  run $M --model mmoecut --num-tasks 2.1 --batch 2048 --steps 3 --warmup 1 --dropout 0.2
  run $M --model mtattncut --num-tasks 3 --buckets 100,200,300 --steps 6 --warmup 3
  run $M --model mtattncut --num-tasks 3 --buckets 100,200,300 --reward dcg --steps 6 --warmup 3
  run $M --model mtattncut --num-tasks 3 --steps 5 --warmup 2
  run $M --model choopy --batch 8192 --steps 2 --warmup 1
  run $M --model choopy --batch 8192 --steps 2 --warmup 1 --dropout 0.2
  run $M --batch 32 --steps 40 --warmup 5
  run $M --batch 63 --steps 40 --warmup 5
  run $M --model choopy --batch 32 --steps 40 --warmup 5
done
cat $T
