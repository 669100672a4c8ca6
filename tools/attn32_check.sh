#!/bin/bash
# GPU box: parity of the exact-fp32 attention kernels and the times of the three launches in a real training step.
TAG=${1:-a32}
R=$GRAFT_REPO_ROOT
python $R/tools/gpu_probe.py attention scale_ops dropout --precision=fp32 > $R/gpurun_out/${TAG}_probe.log 2>&1
grep -n "FAIL\|ok, " $R/gpurun_out/${TAG}_probe.log | tail -4
grep "attn .*B4096" $R/gpurun_out/${TAG}_probe.log
python $R/bench.py --precision fp32 --no-cpu-baseline --other-steps 0 --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_bench.json 2>/dev/null
python -c "
import json; d=json.load(open('$R/gpurun_out/${TAG}_bench.json')); r=d['roofline']; print('fp32 step', d['ms_per_step'], 'dkv', r['launch_ms'], r['frac'], r['other_kernels_ms'])"
