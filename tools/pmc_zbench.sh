#!/bin/bash
# PMC passes over tools/zbench variants (args: binaries under build/); results -> gpurun_out/zpmc_<bin>_<set>/ ; summarise with tools/pmc_zsummary.py
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for bin in "$@"; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
             "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
             "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    out=$R/gpurun_out/zpmc_${bin}_$i
    rm -rf $out
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- $R/build/$bin 8 256 0 10 zonly > /dev/null 2>$out.err
  done
done
