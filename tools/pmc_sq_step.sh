# SQ counters of every kernel of the AttnCut training step (one PMC pass, kernel trace only): clock, matrix-pipe occupancy and the
# wavefront-cycle split (parked at s_waitcnt / barrier, issue-stalled, issuing) per kernel -> gpurun_out/TAG_pmc_sq.txt
export RLT_BENCH_SMALL=0
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/${TAG}_pmc_sq -o sq --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --other-steps 0 ${PMC_SQ_ARGS} > $R/gpurun_out/${TAG}_pmc_sq.log 2>&1 || exit 1
python3 $R/tools/pmc_sq_report.py $R/gpurun_out/${TAG}_pmc_sq "" > $R/gpurun_out/${TAG}_pmc_sq.txt
cat $R/gpurun_out/${TAG}_pmc_sq.txt
