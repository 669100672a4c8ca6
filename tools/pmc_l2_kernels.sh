# L2-side counters of the kernels of one tools/bench_kernels.py benchmark: FETCH_SIZE (fabric-side read requests, KB; doubled for
# wide streaming reads on gfx950) and, in a pass of its own, the L2 hit / miss counts.  bash tools/pmc_l2_kernels.sh TAG BENCH
TAG=${1:-k}
BENCH=${2:-attention_fwd}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -o p -- python3 $R/tools/bench_kernels.py $BENCH > $R/gpurun_out/${TAG}_fetch.log 2>&1 || echo "fetch pass failed"
timeout -k 10 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_tcc -o p -- python3 $R/tools/bench_kernels.py $BENCH > $R/gpurun_out/${TAG}_tcc.log 2>&1 || echo "tcc pass failed"
python3 - <<PY
import csv, collections, glob
for d in ("${TAG}_fetch", "${TAG}_tcc"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$R/gpurun_out/" + d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, {n: round(sum(x) / len(x)) for n, x in v.items()})
PY
