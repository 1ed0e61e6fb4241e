#!/bin/bash
# A/B instruction-count comparison of two tools/kbench builds (rocprofv3 --pmc, separate runs):
#   bash tools/pmc_ab.sh build/kbench_a build/kbench_b      -> gpurun_out/pmc_<name>/p_results.db (summarise like tools/pmc_summary.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for b in "$@"; do
  n=$(basename $b)
  rm -rf $R/gpurun_out/pmc_$n
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --kernel-trace -d $R/gpurun_out/pmc_$n -o p -- $R/$b 8 256 > /dev/null 2>&1
done
