cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in main step2 olddma; do
  if [ $v = main ]; then unset RLT_HIP_LIB; else export RLT_HIP_LIB=$R/ranked-list-truncation_amd/csrc/variants/librlt_$v.so; fi
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_sq_$v -o sq --output-format csv -- python3 $R/tools/bench_kernels.py attention > $R/gpurun_out/pmc_sq_$v.log 2>&1 || exit 1
done
ls $R/gpurun_out/pmc_sq_step2
