#!/bin/bash
# development: build/libtrx_<name>.so with extra -D flags for lncc.hip.  usage: tools/build_variant_lncc.sh name -DX=1 ...
set -e
name=$1; shift
mkdir -p build/v_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -Wno-unused-variable "$@" -c torchregister_amd/csrc/lncc.hip -o build/v_$name/lncc.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libtrx_$name.so build/api.o build/affine.o build/flow.o build/v_$name/lncc.o build/kde.o build/peer.o
