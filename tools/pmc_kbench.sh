#!/bin/bash
set -eu
# PMC passes over tools/kbench (arg 1 = binary); results -> gpurun_out/pmc*/ ; summarise with tools/pmc_summary.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
BIN=${1:-$R/build/kbench4}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc$i
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc$i -o p -- $BIN 8 256 > /dev/null 2>$R/gpurun_out/pmc$i.err
done
