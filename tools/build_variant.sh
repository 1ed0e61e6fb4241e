#!/bin/bash
# development: build/libtrx_<name>.so with extra -D flags for ONE object (affine | flow | lncc; default affine), the other objects being the product's.
#   tools/build_variant.sh [-o flow] name -DX=1 ...      then e.g.  bash tools/bench_variants.sh "name" 2   (alternates libraries on one box)
set -e
obj=affine
if [ "$1" = "-o" ]; then obj=$2; shift 2; fi
name=$1; shift
mkdir -p build/v_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -Wno-unused-variable "$@" -c torchregister_amd/csrc/$obj.hip -o build/v_$name/$obj.o
objs=""
for o in api affine flow lncc kde peer; do if [ $o = $obj ]; then objs="$objs build/v_$name/$o.o"; else objs="$objs build/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libtrx_$name.so $objs
