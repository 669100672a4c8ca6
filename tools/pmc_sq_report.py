"""Per-kernel summary of a rocprofv3 --pmc SQ_* run (counter_collection.csv + kernel_trace.csv in one directory).

usage: python tools/pmc_sq_report.py DIR [kernel-substring]
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, constants table).
"""
import collections
import csv
import glob
import re
import sys


def main():
    d = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "attn3"
    tr = {}
    for r in csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])):
        tr[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(glob.glob(d + "/*counter_collection.csv")[0])):
        name, dur = tr[r["Dispatch_Id"]]
        if want not in name:
            continue
        m = re.search(r"(\w+<[^>]*>)", name)
        k = m.group(1) if m else name[:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["dur"].append(dur)
    for k, c in agg.items():
        avg = {n: sum(v) / len(v) for n, v in c.items()}
        gui = avg.get("GRBM_GUI_ACTIVE", 0) / 8
        wc = avg.get("SQ_WAVE_CYCLES", 0)
        line = f"{k:34s} {avg['dur'] / 1e6:7.3f} ms"
        if gui:
            line += f"  clk {gui / avg['dur']:.2f} GHz  mfma-pipe {avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * gui):.3f}"
        if wc:
            for n, lab in (("SQ_WAIT_ANY", "parked"), ("SQ_WAIT_INST_ANY", "issue-stalled"), ("SQ_ACTIVE_INST_ANY", "issuing"),
                           ("SQ_ACTIVE_INST_VALU", "valu"), ("SQ_ACTIVE_INST_LDS", "lds")):
                if n in avg:
                    line += f"  {lab} {avg[n] / wc:.3f}"
        print(line)


if __name__ == "__main__":
    main()
