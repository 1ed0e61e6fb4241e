#!/bin/bash
# development: tools/bench_lncc.py and the device loop of tools/bench_lncc_loop.py alternately with the product library and build/libtrx_<name>.so
#   bash tools/lncc_variants.sh "lnt1 lnt7" [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
names=$1; rounds=${2:-2}
cp $R/torchregister_amd/lib/libtrx.so /tmp/libtrx_orig.so
for i in $(seq $rounds); do
  for n in base $names; do
    if [ $n = base ]; then cp /tmp/libtrx_orig.so $R/torchregister_amd/lib/libtrx.so; else cp $R/build/libtrx_$n.so $R/torchregister_amd/lib/libtrx.so; fi
    echo "$n: kernels $(python3 $R/tools/bench_lncc.py 2>/dev/null | grep -o '[0-9.]* us' | tr '\n' ' ') | loop $(python3 $R/tools/bench_lncc_loop.py 2>/dev/null | grep -o 'device loop [0-9]* us' | grep -o '[0-9]* us' | tr '\n' ' ')"
  done
done
cp /tmp/libtrx_orig.so $R/torchregister_amd/lib/libtrx.so
