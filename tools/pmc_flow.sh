#!/bin/bash
# Kernel trace + PMC passes over tools/flow_only.py (args passed through); results -> gpurun_out/zpmc_flow_<tag>_<n>/ ; summarise with
#   python3 tools/pmc_zsummary.py flow_ gpurun_out
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  out=$R/gpurun_out/zpmc_flow_${TAG}_$i
  rm -rf $out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- python3 $R/tools/flow_only.py "$@" > $out.log 2>&1
done
