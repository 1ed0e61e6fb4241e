"""Summarise gpurun_out/zpmc_*/ (tools/pmc_zbench.sh): per-launch averages of every counter for kernels matching a pattern."""
import collections, csv, glob, os, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "zstream"
root = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
vox_waves = 8 * 256 ** 3 / 64
for d in sorted(glob.glob(root + "/zpmc_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in acc.items():
            m = sum(v) / len(v)
            print(f"{os.path.basename(d):28s} {c:24s} {m:16.0f}  per voxel-wave {m / vox_waves:9.3f}  (n={len(v)})")
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        du = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
        if du:
            print(f"{os.path.basename(d):28s} {'duration_us':24s} {sum(du) / len(du) / 1e3:16.1f}  (n={len(du)})")
