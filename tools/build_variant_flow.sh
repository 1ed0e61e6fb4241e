#!/bin/bash
# development: build/libtrx_<name>.so with extra -D flags for flow.hip (the other objects are the product's).  usage: tools/build_variant_flow.sh name -DX=1 ...
set -e
name=$1; shift
mkdir -p build/v_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize "$@" -c torchregister_amd/csrc/flow.hip -o build/v_$name/flow.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libtrx_$name.so build/api.o build/affine.o build/v_$name/flow.o build/lncc.o build/kde.o build/peer.o
