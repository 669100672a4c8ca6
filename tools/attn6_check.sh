#!/bin/bash
# GPU box: parity of the bf16x6 attention kernels (attention + scale_ops + models sections at the exact-fp32 tolerances) and
# the times of the three launches in a real training step.  bash tools/attn6_check.sh TAG
TAG=${1:-a6}
R=$GRAFT_REPO_ROOT
python $R/tools/gpu_probe.py attention scale_ops models --precision=bf16x6 > $R/gpurun_out/${TAG}_probe.log 2>&1
grep -n "FAIL\|ok, " $R/gpurun_out/${TAG}_probe.log | tail -8
grep "attn .*B4096" $R/gpurun_out/${TAG}_probe.log
python $R/bench.py --precision bf16x6 --no-cpu-baseline --other-steps 0 --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_bench.json 2>/dev/null
python -c "
import json; d=json.load(open('$R/gpurun_out/${TAG}_bench.json')); r=d['roofline']; print('x6 step', d['ms_per_step'], 'dkv', r['launch_ms'], r['frac'], r['other_kernels_ms'])"
