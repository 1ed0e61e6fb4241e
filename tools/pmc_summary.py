"""Summarise gpurun_out/pmc*/p_results.db (rocprofv3 --pmc over tools/kbench): per-launch counter averages of
affine_tile_kernel<0> for the launches whose box fits (duration < 0.8 ms)."""
import collections, glob, sqlite3, sys
vox_waves = 8 * 256 ** 3 / 64
for f in sorted(glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out") + "/pmc*/p_results.db")):
    db = sqlite3.connect(f)
    rows = db.execute("select dispatch_id, counter_name, value, duration from counters_collection "
                      "where kernel_name like '%affine_tile_kernel<0>%' order by dispatch_id")
    d = collections.OrderedDict()
    for did, c, v, dur in rows:
        d.setdefault(did, {"dur": dur})[c] = v
    g = [x for x in d.values() if x["dur"] < 800000]
    if not g:
        continue
    n = len(g)
    print(f"{f}: {n} launches, avg {sum(x['dur'] for x in g) / n / 1e3:.1f} us")
    for c in g[0]:
        if c != "dur":
            v = sum(x[c] for x in g) / n
            print(f"    {c:28s} {v:16.0f}   per voxel-wave {v / vox_waves:8.2f}")
