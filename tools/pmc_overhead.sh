#!/bin/bash
set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in kbench kb_skip1; do
  rm -rf $R/gpurun_out/pmc_$v
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$v -- $R/build/$v 8 256 > /dev/null 2>$R/gpurun_out/pmc_$v.err
done
ls $R/gpurun_out/pmc_kbench/*/ | head
