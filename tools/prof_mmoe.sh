cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_prof_mmoe -o p -- python3 $R/bench.py --model mmoecut --num-tasks 2.1 --batch 2048 --steps 3 --warmup 1 --no-cpu-baseline --fp32-steps 0 > $O/r02_prof_mmoe.log 2>&1
python3 $R/tools/trace_calls.py $O/r02_prof_mmoe/p_kernel_trace.csv 1 > $O/r02_prof_mmoe_per_call.txt
cat $O/r02_prof_mmoe_per_call.txt
