cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r04_g6c_nostore.log
echo "=== product build" > $O
timeout -k 10 300 python tools/x6_probe.py --modes=bf16x6 >> $O 2>&1
echo "=== no output stores (timing-only ablation)" >> $O
RLT_HIP_LIB=$GRAFT_REPO_ROOT/ranked-list-truncation_amd/csrc/variants/librlt_g6c_nostore.so timeout -k 10 300 python tools/x6_probe.py --modes=bf16x6 >> $O 2>&1
cat $O | cut -c1-150
