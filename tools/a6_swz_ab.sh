# GPU box: parity + same-box A/B of the swizzled LDS layout of the one-wavefront head-dim-64 backward kernels (variant a6_noswz = -DRLT_A6_SWZ=0)
R=$GRAFT_REPO_ROOT; V=$R/ranked-list-truncation_amd/csrc/variants
cd $R
python tools/gpu_probe.py attention scale_attention dropout scale_dropout --precision=bf16x6 2>&1 | tail -3
for i in 1 2; do
echo "=== swizzled 128-byte rows"; python tools/bench_kernels.py attention 2>&1 | grep "attn_bwd_d"
echo "=== padded 144-byte rows"; RLT_HIP_LIB=$V/librlt_a6_noswz.so python tools/bench_kernels.py attention 2>&1 | grep "attn_bwd_d"
done
echo "=== dropout 0.4, swizzled"; python tools/bench_kernels.py attention_drop 2>&1 | grep "attn_bwd_d.*HD64"
echo "=== dropout 0.4, padded"; RLT_HIP_LIB=$V/librlt_a6_noswz.so python tools/bench_kernels.py attention_drop 2>&1 | grep "attn_bwd_d.*HD64"
