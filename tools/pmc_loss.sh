# instruction mix of the loss + cut-metrics pass (tools/bench_kernels.py loss): one PMC pass, kernel trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $R/gpurun_out/pmc_loss -o p --output-format csv -- python3 $R/tools/bench_kernels.py loss > $R/gpurun_out/pmc_loss.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, os, collections
d = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_loss"
tr = {}
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    tr[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["Grid_Size"]) if "Grid_Size" in r else 0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    name, grid = tr[r["Dispatch_Id"]]
    if "reward_loss" not in name and "cut_metrics" not in name:
        continue
    agg[(name[:60], grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    avg = {n: sum(v) / len(v) for n, v in c.items()}
    w = avg.get("SQ_WAVES", 1)
    print(k, "launches", len(c["SQ_WAVES"]), "waves %.0f" % w, " per wave:",
          {n[3:]: round(v / w, 1) for n, v in avg.items() if n != "SQ_WAVES"})
PY
