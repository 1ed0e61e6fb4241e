#!/bin/bash
# development: tools/time_flow_run.py alternately with the product library and build/libtrx_<name>.so variants, several rounds
#   bash tools/flow_variants.sh "w256 w256nt3" [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
names=$1; rounds=${2:-3}
cp $R/torchregister_amd/lib/libtrx.so /tmp/libtrx_orig.so
for i in $(seq $rounds); do
  for n in base $names; do
    if [ $n = base ]; then cp /tmp/libtrx_orig.so $R/torchregister_amd/lib/libtrx.so; else cp $R/build/libtrx_$n.so $R/torchregister_amd/lib/libtrx.so; fi
    echo "$n: $(python3 $R/tools/time_flow_run.py 2>/dev/null | grep -o '[0-9.]* us' | tr '\n' ' ')"
  done
done
cp /tmp/libtrx_orig.so $R/torchregister_amd/lib/libtrx.so
