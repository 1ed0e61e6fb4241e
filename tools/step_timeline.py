#!/usr/bin/env python3
"""Timeline of an affine step from a rocprofv3 --kernel-trace CSV: for every kernel of the step its duration and the gap to the kernel in
front of it on the device (start - previous end), averaged over the steady part of the trace.
    python tools/step_timeline.py <dir with *_kernel_trace.csv> [first_step last_step]"""
import csv, glob, os, sys
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: n.replace("void trx::", "").split("(")[0]
# a step starts at each zs_step (or dual / accum) kernel: group until the next finalize
steps, cur = [], []
for s, e, n in rows:
    n = short(n)
    if not n.startswith("affine_"):
        continue
    cur.append((s, e, n))
    if "finalize" in n:
        steps.append(cur); cur = []
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(steps) // 4
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(steps) - 2
sig = {}
for i in range(max(lo, 1), hi):
    st, prev_end = steps[i], steps[i - 1][-1][1]
    key = tuple(n for _, _, n in st)
    acc = sig.setdefault(key, {"n": 0, "dur": [0.0] * len(st), "gap": [0.0] * len(st), "span": 0.0})
    acc["n"] += 1
    for k, (s, e, n) in enumerate(st):
        acc["dur"][k] += (e - s) * 1e-3
        acc["gap"][k] += (s - prev_end) * 1e-3
        prev_end = e
    acc["span"] += (st[-1][1] - steps[i - 1][-1][1]) * 1e-3
for key, acc in sorted(sig.items(), key=lambda kv: -kv[1]["n"]):
    n = acc["n"]
    print(f"-- {n} steps of {len(key)} kernels, {acc['span'] / n:.2f} us from the end of one step to the end of the next")
    for k, name in enumerate(key):
        print(f"   {name:44s} gap {acc['gap'][k] / n:7.2f} us   duration {acc['dur'][k] / n:8.2f} us")
