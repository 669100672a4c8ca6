# LDS counters of the kernels of one tools/bench_kernels.py benchmark: bank-conflict cycles as a share of all LDS-array cycles.
# bash tools/pmc_lds_kernels.sh TAG BENCH
TAG=${1:-k}
BENCH=${2:-attention}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_lds -o p -- python3 $R/tools/bench_kernels.py $BENCH > $R/gpurun_out/${TAG}_lds.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/${TAG}_lds/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {n: sum(x) / len(x) for n, x in v.items()}
    if m.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        print(f"{k:72s} LDS cycles {m['SQ_LDS_IDX_ACTIVE']:.3e}  bank-conflict cycles {m.get('SQ_LDS_BANK_CONFLICT', 0):.3e}  share {m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:.4f}")
PY
