#!/bin/bash
# HBM-side traffic, L2 and SQ counters of the F1 launch at the two rotated bench poses (VERDICT r3 #1a) -> gpurun_out/zpmc_pose_<pose>_<n>;
# summarise with  python3 tools/pmc_zsummary.py affine_tile_dual gpurun_out
#   bash tools/pmc_pose.sh [flags]
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
FL=${1:-0}
for pose in rot rigid; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
             "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"; do
    i=$((i+1)); out=$R/gpurun_out/zpmc_pose_${pose}_$i; rm -rf $out
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- python3 $R/tools/f1_at_pose.py $pose 20 $FL > $out.log 2>&1
  done
done
