#!/bin/bash
# Round-6 evidence set on one box, one library: PMC + kernel stats of the bench command (-> traffic.json), PMC of the rotated pose
# (-> traffic_rot.json) and of the flow loop (-> traffic_flow.json), then the bench line that quotes all three, the rotation sweep,
# the BASELINE configs, the launch-bound sizes and the one-kernel A/B.     bash tools/profile_r06.sh <tag> "<note>"
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
rm -rf $O && mkdir -p $O
cd $R
bash tools/profile_bench.sh > $O/profile_bench.log 2>&1
python3 tools/summarize_prof.py gpurun_out/prof $1 "$2" > $O/summary.log 2>&1
bash tools/pmc_pose.sh 0 > $O/pmc_pose.log 2>&1
{ echo "== affine_zs_step_kernel (takes no pair at this pose)"; python3 tools/pmc_zsummary.py affine_zs_step gpurun_out 2>/dev/null | grep zpmc_pose; echo "== affine_eft_step_kernel"; python3 tools/pmc_zsummary.py affine_eft_step gpurun_out 2>/dev/null | grep zpmc_pose; echo "== affine_tile_dual_kernel (skips every pair)"; python3 tools/pmc_zsummary.py affine_tile_dual gpurun_out 2>/dev/null | grep zpmc_pose; } > $O/pose_pmc.txt
bash tools/pmc_flow.sh final adam 1.0 30 > /dev/null 2>&1
python3 tools/pmc_zsummary.py flow_ gpurun_out 2>/dev/null | grep zpmc_flow_final > $O/flow_pmc.txt
python3 tools/summarize_traffic.py $1 > $O/traffic_summary.log 2>&1
cp profiles/traffic.json profiles/traffic_rot.json profiles/traffic_flow.json $O/ 2>/dev/null; cp profiles/$1_bench_rocprof_summary.csv $O/
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_steps20.json 2>/dev/null
python3 tools/bench_rotation.py 0 2>&1 | grep -v amdgpu > $O/rotation_sweep.txt
python3 tools/bench_configs.py 2>&1 | grep -v amdgpu > $O/configs.txt
python3 tools/bench_small.py 2>&1 | grep -v amdgpu >> $O/configs.txt
python3 tools/bench_lncc_loop.py 2>&1 | grep -v amdgpu > $O/lncc_loop.txt
rm -rf gpurun_out/prof gpurun_out/zpmc_*
