#!/bin/bash
# Round-3 evidence set on one box, one library: bench line, rocprofv3 kernel stats + PMC passes of the bench command (-> traffic.json),
# flow kernels (kernel trace + FETCH / WRITE), rotation sweep, BASELINE configs, local-NCC loop, z-streaming A/B.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
rm -rf $O && mkdir -p $O
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err
bash tools/profile_bench.sh > $O/profile_bench.log 2>&1
python3 tools/summarize_prof.py gpurun_out/prof $1 "$2" > $O/summary.log 2>&1
cp profiles/traffic.json $O/traffic.json; cp profiles/$1_bench_rocprof_summary.csv $O/
python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_steps20.json 2>/dev/null
python3 tools/bench_rotation.py 0 2>&1 | grep -v amdgpu > $O/rotation_sweep.txt
python3 tools/bench_configs.py 2>&1 | grep -v amdgpu > $O/configs.txt
python3 tools/bench_lncc_loop.py 2>&1 | grep -v amdgpu > $O/lncc_loop.txt
python3 tools/bench_zs_ab.py 8 256 2>&1 | grep -v amdgpu > $O/zs_ab.txt
python3 tools/bench_small.py 2>&1 | grep -v amdgpu > $O/small.txt
for cfg in "adam 1.0 30" "adam 0.0 30" "sgd 0.0 30" "adam 1.0 30 64 512 512"; do
  tag=$(echo $cfg | tr ' .' '__')
  bash tools/pmc_flow.sh $tag $cfg > /dev/null 2>&1
done
python3 tools/pmc_zsummary.py flow_ gpurun_out 2>/dev/null | grep zpmc_flow > $O/flow_pmc_all.txt
for k in flow_update3 flow_moments3 flow_coef; do python3 tools/pmc_zsummary.py $k gpurun_out 2>/dev/null | grep zpmc_flow | sed "s/^/$k /" ; done > $O/flow_pmc_by_kernel.txt
