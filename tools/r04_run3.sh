cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r04_run3.log
timeout -k 10 600 python tools/gpu_probe.py --precision=bf16x6 lstm losses scale_ops full_size_kernels models trajectory > $O 2>&1
echo "exit $?" >> $O
grep -n "^FAIL\|^--- \|ok, .* failed\|Traceback\|Error" $O | head -40
timeout -k 10 600 python bench.py > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err
echo "bench exit $?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_a.json').read().strip().splitlines()[-1])
print(d['precision_mode'], d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['other_kernels_ms'])
for k in ('f32_mfma_mode','fast_mode'):
    if k in d: print(k, d[k]['ms_per_step'], d[k]['roofline']['frac'], d[k]['roofline']['other_kernels_ms'])
print('hbm', d['hbm_kernel']['achieved'], d['hbm_kernel']['frac'], d['hbm_kernel']['stream_reference'])
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
