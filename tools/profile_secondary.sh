#!/bin/bash
# Kernel statistics of the secondary paths (flow steps, warps, local NCC, rigid steps: tools/bench_misc.py; default criterion: tools/bench_default_criterion.py)
# -> gpurun_out/sec/{misc,defcrit}; condense with:  python tools/summarize_secondary.py <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/sec && mkdir -p $R/gpurun_out/sec
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sec/misc -- python3 $R/tools/bench_misc.py > $R/gpurun_out/sec/misc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sec/defcrit -- python3 $R/tools/bench_default_criterion.py > $R/gpurun_out/sec/defcrit.log 2>&1
grep -v "^[WEI]2026\|amdgpu.ids" $R/gpurun_out/sec/misc.log | tail -25
grep -v "^[WEI]2026\|amdgpu.ids" $R/gpurun_out/sec/defcrit.log | tail -2
