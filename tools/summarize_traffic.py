#!/usr/bin/env python3
"""profiles/traffic_rot.json and profiles/traffic_flow.json from the PMC passes of tools/pmc_pose.sh (rot pose) and tools/pmc_flow.sh
(tag final: adam 1.0) under gpurun_out/: HBM-side bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950 tallies 128-byte read requests
at 64 bytes: MI355X_MICROARCH.md, HBM section; calibrated there for wide coalesced reads - the flow kernels read one dword per lane, for which
the factor is the guide's stated assumption, not a calibration), summed over the kernels of one F1 step / one flow iteration, with the
library's hash so that bench.py quotes them only for the library they were measured on.
    python3 tools/summarize_traffic.py <tag>"""
import collections, csv, glob, hashlib, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
sha = hashlib.sha256(open(os.path.join(root, "torchregister_amd", "lib", "libtrx.so"), "rb").read()).hexdigest()

def counters(pattern_dir, kernels):
    """{counter: sum over the kernels of the per-launch average}, launches per kernel"""
    out, launches = collections.defaultdict(float), {}
    for d in sorted(glob.glob(os.path.join(root, "gpurun_out", pattern_dir))):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                for k in kernels:
                    if k in r["Kernel_Name"]:
                        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, cs in acc.items():
                for c, v in cs.items():
                    out[c] += sum(v) / len(v)
                    launches[k] = len(v)
    return out, launches

# (the step kernels only - "<0": the one MODE 3 launch of affine_tile_dual_kernel that builds the synthetic moving volumes is not part of a step)
rot, n = counters("zpmc_pose_rot_*", ("affine_zs_step_kernel<0", "affine_eft_step_kernel<0", "affine_tile_dual_kernel<0"))
if "FETCH_SIZE" in rot and "WRITE_SIZE" in rot:
    j = {"what": "one F1 step of 8 x 256^3 at theta = R(0.5,0.4,0.3) diag(1.05,0.95,1.02): affine_zs_step_kernel<0> (takes no pair) + affine_eft_step_kernel<0> + affine_tile_dual_kernel<0,4> (which skips every pair)",
         "FETCH_SIZE_KiB": rot["FETCH_SIZE"], "WRITE_SIZE_KiB": rot["WRITE_SIZE"], "hbm_bytes_per_launch": 2 * rot["FETCH_SIZE"] * 1024 + rot["WRITE_SIZE"] * 1024,
         "algorithmic_bytes_per_launch": 8 * 256 ** 3 * 8, "l2_requests_per_launch": rot.get("TCC_REQ_sum"), "l2_misses_per_launch": rot.get("TCC_MISS_sum"),
         "launches_sampled": n, "tag": tag, "lib_sha256": sha}
    json.dump(j, open(os.path.join(root, "profiles", "traffic_rot.json"), "w"), indent=1)
    print("rot: HBM-side GB per step", j["hbm_bytes_per_launch"] / 1e9, "= x", j["hbm_bytes_per_launch"] / j["algorithmic_bytes_per_launch"])
fl, n = counters("zpmc_flow_final_*", ("flow_update3_kernel", "flow_coef_kernel"))
if "FETCH_SIZE" in fl and "WRITE_SIZE" in fl:
    j = {"what": "one iteration of trx_flow_run, 1 x 256^3, Adam + smoothness: flow_update3_kernel + flow_coef_kernel",
         "FETCH_SIZE_KiB": fl["FETCH_SIZE"], "WRITE_SIZE_KiB": fl["WRITE_SIZE"], "hbm_bytes_per_iteration": 2 * fl["FETCH_SIZE"] * 1024 + fl["WRITE_SIZE"] * 1024,
         "algorithmic_bytes_per_iteration": 80 * 256 ** 3, "l2_requests_per_iteration": fl.get("TCC_REQ_sum"), "launches_sampled": n, "tag": tag, "lib_sha256": sha,
         "note": "FETCH_SIZE x 2 is calibrated for 16-byte-per-lane reads; these kernels read one dword per lane"}
    json.dump(j, open(os.path.join(root, "profiles", "traffic_flow.json"), "w"), indent=1)
    print("flow: HBM-side GB per iteration", j["hbm_bytes_per_iteration"] / 1e9, "= x", j["hbm_bytes_per_iteration"] / j["algorithmic_bytes_per_iteration"])
