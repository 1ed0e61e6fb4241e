#!/usr/bin/env python3
"""gpurun_out/sec/{misc,defcrit} (tools/profile_secondary.sh) -> profiles/<tag>_secondary_kernels.csv (the trx:: kernels only)."""
import csv, glob, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = ["# rocprofv3 --kernel-trace --stats -- python3 tools/bench_misc.py / tools/bench_default_criterion.py   (MI355X, 256^3 resp. 128^3, 1 pair unless "
       "noted in the tool; averages include warm-up launches)", "Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs"]
for sub, title in (("misc", "bench_misc"), ("defcrit", "bench_default_criterion 128")):
    fs = sorted(glob.glob(os.path.join(root, "gpurun_out", "sec", sub, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if not fs:
        continue
    out.append(f"# ---- {title}")
    for r in csv.DictReader(open(fs[-1])):
        if "trx::" in r["Name"]:
            out.append(",".join([r["Name"].split("(")[0].replace(",", ";"), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]]))
open(os.path.join(root, "profiles", f"{tag}_secondary_kernels.csv"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
