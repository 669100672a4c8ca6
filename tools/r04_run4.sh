cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/gpu_probe.py --precision=bf16x6 lstm scale_ops full_size_kernels 2>&1 | grep -n "^FAIL\|^--- \|ok, .* failed\|Traceback\|Error" | head
timeout -k 10 300 python bench.py --no-cpu-baseline --other-steps 0 > gpurun_out/r04_bench_b.json 2> gpurun_out/r04_bench_b.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_b.json').read().strip().splitlines()[-1])
print(d['precision_mode'], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['other_kernels_ms'])
PY
RLT_LSTM6=0 timeout -k 10 300 python bench.py --no-cpu-baseline --other-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RLT_LSTM6=0', d['ms_per_step'], d['roofline']['other_kernels_ms'])"
