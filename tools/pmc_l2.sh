#!/bin/bash
set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 -L 2>/dev/null | grep -oE "TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+" | sort -u > $R/gpurun_out/tcc_counters.txt
i=10
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" ; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc$i
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc$i -o p -- $R/build/kbench16 8 256 > /dev/null 2>$R/gpurun_out/pmc$i.err
  tail -1 $R/gpurun_out/pmc$i.err | cut -c1-200
done
