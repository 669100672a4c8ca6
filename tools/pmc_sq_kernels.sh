# SQ counters of the kernels of one tools/bench_kernels.py benchmark (one PMC pass, kernel trace only) -> gpurun_out/TAG_pmc_sq.txt
# bash tools/pmc_sq_kernels.sh TAG BENCH [kernel-substring]     (RLT_HIP_LIB selects a variant library)
TAG=${1:-k}
BENCH=${2:-attention}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/${TAG}_pmc_sq -o sq --output-format csv -- python3 $R/tools/bench_kernels.py $BENCH > $R/gpurun_out/${TAG}_pmc_sq.log 2>&1 || exit 1
python3 $R/tools/pmc_sq_report.py $R/gpurun_out/${TAG}_pmc_sq "$3" > $R/gpurun_out/${TAG}_pmc_sq.txt
cat $R/gpurun_out/${TAG}_pmc_sq.txt
