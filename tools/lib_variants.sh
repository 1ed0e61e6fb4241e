#!/bin/bash
# development: run a command once per library variant build/libtrx_<name>.so (swapped in for the product library on the GPU box's copy)
#   bash tools/lib_variants.sh "big0 skip0" python tests/fuzz_affine.py 200 5 79
R=${GRAFT_REPO_ROOT:-/root/repo}
names=$1; shift
cp $R/torchregister_amd/lib/libtrx.so /tmp/libtrx_orig.so
for n in $names; do
  cp $R/build/libtrx_$n.so $R/torchregister_amd/lib/libtrx.so
  echo "== variant $n"; "$@" 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300
done
cp /tmp/libtrx_orig.so $R/torchregister_amd/lib/libtrx.so
