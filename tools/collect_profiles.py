#!/usr/bin/env python3
"""After `gpurun -- 'bash tools/refresh_profiles.sh TAG'`: copy the summaries to be judged from gpurun_out/ (scratch) into
profiles/ (tracked): the bench line, per mode the rocprofv3 kernel-stats CSV, the per-call table (tools/trace_calls.py) and
the PMC traffic JSON (tools/pmc_traffic.py), and the Choopy stats.

    python tools/collect_profiles.py r03
"""
import glob
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(REPO, "gpurun_out"), os.path.join(REPO, "profiles")


def one(pattern):
    hits = sorted(glob.glob(os.path.join(O, pattern), recursive=True))
    return hits[0] if hits else None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    bench = os.path.join(O, f"{tag}_bench_n1.json")
    if os.path.exists(bench):
        shutil.copy(bench, os.path.join(P, f"{tag}_bench_n1.json"))
    for mode in ("fp32", "bf16x3", "bf16x6"):
        stats = one(f"{tag}_prof_attncut_{mode}/**/p_kernel_stats.csv")
        trace = one(f"{tag}_prof_attncut_{mode}/**/p_kernel_trace.csv")
        if stats:
            shutil.copy(stats, os.path.join(P, f"{tag}_{mode}_attncut_b4096_kernel_stats.csv"))
        if trace:
            with open(os.path.join(P, f"{tag}_{mode}_attncut_b4096_per_call.txt"), "w") as f:
                subprocess.run([sys.executable, os.path.join(REPO, "tools", "trace_calls.py"), trace], stdout=f, check=True)
        fetch = one(f"{tag}_pmc_fetch_{mode}/**/p_counter_collection.csv")
        write = one(f"{tag}_pmc_write_{mode}/**/p_counter_collection.csv")
        if fetch and write:
            dom = {"fp32": "attn_bwd_dkv_kernel", "bf16x3": "attn3_bwd_dkv_kernel", "bf16x6": "attn6_bwd_dkv"}[mode]      # (prefix: attn6_bwd_dkv1_kernel)
            with open(os.path.join(P, f"{tag}_pmc_traffic_{mode}.json"), "w") as f:
                # steps in the profiled command: 1 warm-up + 2 timed + 3 kernel-timing steps
                subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_traffic.py"), fetch, write, dom,
                                f"attncut b4096 s300 {mode}", "6"], stdout=f, check=True)
    stats = one(f"{tag}_prof_choopy/**/p_kernel_stats.csv")
    trace = one(f"{tag}_prof_choopy/**/p_kernel_trace.csv")
    if stats:
        shutil.copy(stats, os.path.join(P, f"{tag}_bench_choopy_b8192_kernel_stats.csv"))
    if trace:
        with open(os.path.join(P, f"{tag}_bench_choopy_b8192_per_call.txt"), "w") as f:
            subprocess.run([sys.executable, os.path.join(REPO, "tools", "trace_calls.py"), trace], stdout=f, check=True)
    # the default-mode Choopy profile (tools/choopy_profile.sh), the SQ counter reports and the side configurations
    for name in (f"{tag}_choopy_bf16x6_per_call.txt", f"{tag}_choopy_bf16x6_kernel_stats.csv", f"{tag}_pmc_sq_choopy_bf16x6.txt",
                 f"{tag}_pmc_traffic_choopy_bf16x6.json", f"{tag}_pmc_sq_bf16x6.txt", f"{tag}_pmc_sq_fp32.txt", f"{tag}_pmc_sq_bf16x3.txt",
                 f"{tag}_side_configs.txt"):
        src = os.path.join(O, name)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(P, name))
    for f in sorted(os.listdir(P)):
        if f.startswith(tag):
            print(f)


if __name__ == "__main__":
    main()
