#!/bin/bash
# Runs on the GPU box: BASELINE configs[2] (Choopy, 8192 lists x 300, head dim 16) in the DEFAULT bf16x6 mode - rocprofv3 kernel
# stats + per-call table, the SQ counter pass and the two PMC traffic passes -> gpurun_out/TAG_choopy_bf16x6_* 
set -e
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
ARGS="--precision bf16x6 --model choopy --batch 8192 --steps 2 --warmup 1 --no-cpu-baseline --other-steps 0"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_choopy_bf16x6 -o p -- python3 $R/bench.py $ARGS > $O/${TAG}_prof_choopy_bf16x6.log 2>&1
python3 $R/tools/trace_calls.py $(find $O/${TAG}_prof_choopy_bf16x6 -name p_kernel_trace.csv | head -1) > $O/${TAG}_choopy_bf16x6_per_call.txt
cp $(find $O/${TAG}_prof_choopy_bf16x6 -name p_kernel_stats.csv | head -1) $O/${TAG}_choopy_bf16x6_kernel_stats.csv
head -16 $O/${TAG}_choopy_bf16x6_per_call.txt
if [ "$2" != "statsonly" ]; then
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $O/${TAG}_pmc_sq_choopy -o sq --output-format csv -- python3 $R/bench.py $ARGS > $O/${TAG}_pmc_sq_choopy.log 2>&1
python3 $R/tools/pmc_sq_report.py $O/${TAG}_pmc_sq_choopy "" > $O/${TAG}_pmc_sq_choopy_bf16x6.txt
head -20 $O/${TAG}_pmc_sq_choopy_bf16x6.txt
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch_choopy -o p -- python3 $R/bench.py $ARGS > $O/${TAG}_pmc_fetch_choopy.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_write_choopy -o p -- python3 $R/bench.py $ARGS > $O/${TAG}_pmc_write_choopy.log 2>&1
python3 $R/tools/pmc_traffic.py $(find $O/${TAG}_pmc_fetch_choopy -name p_counter_collection.csv | head -1) $(find $O/${TAG}_pmc_write_choopy -name p_counter_collection.csv | head -1) attn6n_bwd1_kernel "choopy b8192 s300 bf16x6" 6 > $O/${TAG}_pmc_traffic_choopy_bf16x6.json
head -c 600 $O/${TAG}_pmc_traffic_choopy_bf16x6.json
fi
