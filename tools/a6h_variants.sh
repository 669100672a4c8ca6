R=$GRAFT_REPO_ROOT; V=$R/ranked-list-truncation_amd/csrc/variants
for i in 1 2; do
for L in "" h_nodma h_dma24; do
  if [ -z "$L" ]; then echo "--- product"; python $R/tools/bench_kernels.py attention_fwd; else echo "--- $L"; RLT_HIP_LIB=$V/librlt_$L.so python $R/tools/bench_kernels.py attention_fwd; fi
done; done 2>&1 | grep -v "amdgpu.ids\|^env"
