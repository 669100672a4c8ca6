cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_a16 -o sq --output-format csv -- python3 $R/tools/bench_kernels.py attention16 attention > $R/gpurun_out/pmc_a16.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS SQ_WAVES --kernel-trace -d $R/gpurun_out/pmc_a16b -o sq --output-format csv -- python3 $R/tools/bench_kernels.py attention16 attention > $R/gpurun_out/pmc_a16b.log 2>&1
ls $R/gpurun_out/pmc_a16b
