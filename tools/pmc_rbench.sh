#!/bin/bash
# PMC passes over build/rbench (the shelved packed-box kernels) at one pose: bash tools/pmc_rbench.sh <tag> <ax> <ay> <az>; summarise with
#   python3 tools/pmc_zsummary.py affine_rot gpurun_out
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  out=$R/gpurun_out/zpmc_rbench_${TAG}_$i
  rm -rf $out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- $R/build/rbench 8 256 3 "$@" > $out.log 2>&1
done
