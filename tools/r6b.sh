#!/bin/bash
# round 6, call b: timeline of a step, pair-group-major schedule (time, power, FETCH_SIZE) with default-policy target loads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6b; mkdir -p $O; cd $R
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --no-cpu-baseline --no-pose-legs --steps 60 --warmup 20 > $O/trace_bench.json 2> $O/trace_bench.err
python3 tools/step_timeline.py $O/trace > $O/step_timeline.txt 2>&1
rm -rf $O/trace
cp torchregister_amd/lib/libtrx.so /tmp/libtrx_orig.so
for lib in base tgtdef; do
  if [ $lib = base ]; then cp /tmp/libtrx_orig.so torchregister_amd/lib/libtrx.so; else cp build/libtrx_$lib.so torchregister_amd/lib/libtrx.so; fi
  for g in 8 2; do for K in 1 48; do
    echo "== $lib g $g K $K" >> $O/group_power.txt
    bash tools/power_any.sh python3 tools/probe_group_fetch.py $g $K 19200 2>&1 | grep -v amdgpu.ids >> $O/group_power.txt
  done; done
  for g in 8 2; do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${lib}_g$g -o p -- python3 tools/probe_group_fetch.py $g 48 96 > /dev/null 2>&1
    python3 - <<PY >> $O/group_fetch.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/pmc_${lib}_g$g/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "affine_zs_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v = v[len(v) // 4:]
    print("$lib group $g K 48:", k, "launches", len(v), "FETCH_SIZE KiB per launch", sum(v) / len(v), "-> 2 x FETCH bytes per pair-iteration / algorithmic", 2 * 1024 * sum(v) / len(v) / $g / (8 * 256 ** 3))
PY
    rm -rf $O/pmc_${lib}_g$g
  done
done
cp /tmp/libtrx_orig.so torchregister_amd/lib/libtrx.so
cat $O/step_timeline.txt $O/group_power.txt $O/group_fetch.txt
