#!/usr/bin/env python3
"""Per-call view of a rocprofv3 --kernel-trace CSV: the dispatches of one training step in launch order, the same
kernel at different grid sizes kept apart (the stats CSV merges them), averaged over the steps in the trace.

    python tools/trace_calls.py <..._kernel_trace.csv> [steps_to_skip]
"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    path = sys.argv[1]
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                         int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"])),
                         int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count", 0) or 0), int(r["LDS_Block_Size"])))
    rows.sort()
    acc = collections.OrderedDict()
    for s, e, name, wgs, vg, lds in rows:
        k = (name, wgs)
        a = acc.setdefault(k, [0, 0.0, vg, lds, s])
        a[0] += 1
        a[1] += (e - s) / 1e6
    total = sum(a[1] for a in acc.values())
    print(f"{'kernel':72s} {'WGs':>8s} {'calls':>6s} {'avg ms':>9s} {'sum ms':>9s} {'%':>6s} {'regs':>5s} {'LDS':>7s}")
    for (name, wgs), (n, ms, vg, lds, _) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        if ms / total < 0.001:
            continue
        print(f"{name:72s} {wgs:8d} {n:6d} {ms / n:9.3f} {ms:9.2f} {100 * ms / total:6.2f} {vg:5d} {lds:7d}")
    print(f"total kernel time {total:.1f} ms over {len(rows)} dispatches")


if __name__ == "__main__":
    main()
