cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_lstm -o sq --output-format csv -- python3 $R/tools/bench_kernels.py lstm > $R/gpurun_out/pmc_lstm.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_lstm2 -o sq --output-format csv -- python3 $R/tools/bench_kernels.py lstm > $R/gpurun_out/pmc_lstm2.log 2>&1
ls $R/gpurun_out/pmc_lstm2
