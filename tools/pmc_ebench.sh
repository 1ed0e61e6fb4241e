#!/bin/bash
# PMC passes over build/<bin> (tools/ebench.hip variants) at one pose: bash tools/pmc_ebench.sh <bin> <tag> <ax> <ay> <az>; summarise with
#   python3 tools/pmc_zsummary.py affine_eft gpurun_out
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
BIN=$1; TAG=$2; shift; shift
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "FETCH_SIZE"; do
  i=$((i+1))
  out=$R/gpurun_out/zpmc_${TAG}_$i
  rm -rf $out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- $R/build/$BIN 8 256 "$@" 1.0 10 only > $out.log 2>&1
done
