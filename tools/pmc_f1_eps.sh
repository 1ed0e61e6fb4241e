#!/bin/bash
# FETCH_SIZE / TCC counters of the F1 launch at several distances of theta from the identity -> gpurun_out/zpmc_eps<e>_<n>
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for e in "$@"; do
  i=0
  for set in "FETCH_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1)); out=$R/gpurun_out/zpmc_eps${e}_$i; rm -rf $out
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- python3 $R/tools/f1_at_eps.py $e > $out.log 2>&1
  done
done
