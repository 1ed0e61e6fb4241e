#!/usr/bin/env python3
"""Condense a gpurun_out/prof/{stats,fetch,write,sq} rocprofv3 output tree into profiles/<tag>_*.csv + traffic.json."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-pose-legs  (MI355X; {note})",
       "Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
rows = list(csv.DictReader(open(sorted(glob.glob(f"{src}/stats/*/*_kernel_stats.csv"), key=os.path.getmtime)[-1])))
for r in rows:
    if "trx::" in r["Name"]:
        out.append(",".join([r["Name"].split("(")[0].replace(",", ";"), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]))
oth = sum(float(r["TotalDurationNs"]) for r in rows if "trx::" not in r["Name"])
out.append(f"(torch kernels: synthetic-input generation and checks),,{oth:.0f},,,,")
res = {}
out += ["", "# PMC passes (separate runs with --pmc only): average per dispatch", "Kernel,Counter,AvgPerDispatch,Dispatches"]
for name in ("fetch", "write", "sq", "sqw", "l2"):
    fs = sorted(glob.glob(f"{src}/{name}/*/*_counter_collection.csv"), key=os.path.getmtime)[-1:]
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "trx::" in k:
            for c, x in sorted(v.items()):
                out.append(f"{k.replace(',', ';')},{c},{sum(x) / len(x):.1f},{len(x)}")
                res[(k, c)] = sum(x) / len(x)
open(os.path.join(root, "profiles", f"{tag}_bench_rocprof_summary.csv"), "w").write("\n".join(out) + "\n")
dom = [k for k in {k for k, _ in res} if "affine_tile" in k or "affine_accum" in k or "affine_zs_step" in k]
if dom:
    # the F1 step kernel (MODE 0), not the warp that builds the synthetic inputs; since round 5 the z-streaming kernel in front of the tile kernel
    k = max(dom, key=lambda n: ("affine_zs_step_kernel<0" in n, "dual_kernel<0" in n or n.endswith("tile_kernel<0>"), "dual" in n, n))
    fetch, write = res.get((k, "FETCH_SIZE")), res.get((k, "WRITE_SIZE"))
    if fetch is not None and write is not None:
        traffic = 2 * fetch * 1024 + write * 1024
        json.dump({"kernel": k, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                   "correction": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024: gfx950 tallies 128-B read requests at 64 B "
                                 "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
                   "hbm_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": 8 * 256 ** 3 * 8,
                   "l2_requests_per_launch": res.get((k, "TCC_REQ_sum")), "l2_hits_per_launch": res.get((k, "TCC_HIT_sum")),
                   "l2_misses_per_launch": res.get((k, "TCC_MISS_sum")), "l2_request_bytes": 128, "tag": tag,
                   "lib_sha256": hashlib.sha256(open(os.path.join(root, "torchregister_amd", "lib", "libtrx.so"), "rb").read()).hexdigest()},
                  open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
        print("traffic GB/launch", traffic / 1e9)
print("\n".join(out))
