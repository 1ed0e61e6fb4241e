#!/bin/bash
# End-of-round evidence on one box with the final library: rocprofv3 kernel stats + PMC passes of the bench command (-> profiles/traffic.json with
# the library's hash), then the bench line itself (so that it quotes that traffic), and the numbers that moved late in the round.
#   bash tools/profile_final.sh <tag> "<note>"       (results under gpurun_out/<tag>/)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
rm -rf $O && mkdir -p $O
cd $R
bash tools/profile_bench.sh > $O/profile_bench.log 2>&1
python3 tools/summarize_prof.py gpurun_out/prof $1 "$2" > $O/summary.log 2>&1
cp profiles/traffic.json $O/traffic.json; cp profiles/$1_bench_rocprof_summary.csv $O/
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_steps20.json 2>/dev/null
python3 tools/bench_lncc.py 2>&1 | grep -v amdgpu > $O/lncc.txt
python3 tools/bench_lncc_loop.py 2>&1 | grep -v amdgpu > $O/lncc_loop.txt
python3 tools/bench_configs.py 2>&1 | grep -v amdgpu > $O/configs.txt
