#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).

On the GPU box:
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
then
  python tools/pmc_traffic.py gpurun_out/pmc_fetch/p_counter_collection.csv gpurun_out/pmc_write/p_counter_collection.csv \
         attn3_bwd_dkv_kernel "attncut b4096 s300 bf16x3" 6 > profiles/r02_pmc_traffic.json   # 6 = warm-up + steps + the 3 kernel-timing steps

Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md: both counters are in KB; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is exact."""
import collections
import csv
import json
import re
import sys

KERNEL_RE = re.compile(r"(attn(?:3|6n|6h|6|16)?_\w+|gemm6[bces]?_kernel<[^>]*>|gemm3?b?_kernel<[^>]*>|bilstm(?:3|6w|6)?_\w+|splitk_reduce_kernel|add_ln_\w+|heads_\w+|"
                       r"narrow_dw_\w+|reward_loss_kernel|reward_loss_h_kernel|loss_metrics_final_kernel|adam_kernel|rlt_rows_reduce_kernel|colsum_\w+|"
                       r"cut_metrics_kernel|embed_\w+|mmoe_\w+|pair_softmax_\w+|mt_\w+_kernel|dropout_mask_kernel)")


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            m = KERNEL_RE.search(r["Kernel_Name"])
            acc[m.group(1) if m else "other"].append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "mean_kb": sum(v) / len(v)} for k, v in acc.items()}


def main():
    fetch_csv, write_csv, dominant, workload = sys.argv[1:5]
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 3          # bench steps + warm-ups in the profiled command
    fetch = per_kernel(fetch_csv, "FETCH_SIZE")
    write = per_kernel(write_csv, "WRITE_SIZE")
    fk = [k for k in fetch if k.startswith(dominant)]
    wk = [k for k in write if k.startswith(dominant)]
    if not fk or not wk:
        sys.exit(f"kernel {dominant} not found in the counter files")
    read_b = 2.0 * 1024.0 * sum(fetch[k]["mean_kb"] for k in fk)
    write_b = 1024.0 * sum(write[k]["mean_kb"] for k in wk)
    step_b = (sum(2.0 * 1024.0 * v["mean_kb"] * v["launches"] for v in fetch.values())
              + sum(1024.0 * v["mean_kb"] * v["launches"] for v in write.values())) / steps
    json.dump({"workload": workload, "dominant_launch": fk, "step_traffic_bytes": step_b, "steps_profiled": steps,
               "read_bytes_per_launch": read_b,
               "write_bytes_per_launch": write_b, "traffic_bytes_per_launch": read_b + write_b,
               "note": "FETCH_SIZE (KB) doubled per the gfx950 correction, WRITE_SIZE (KB) as is; separate --pmc passes",
               "counters": {"FETCH_SIZE": fetch, "WRITE_SIZE": write}}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
