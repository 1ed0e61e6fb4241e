#!/bin/bash
# builds build/sbench and its ablation variants build/sbench_d<bits> (TRX_SB_DBG); extra hipcc flags after the variant list: tools/sb_build.sh "0 2 4 8" -DX=1
set -e
cd "$(dirname "$0")/.."
mkdir -p build
V=${1:-0}; shift || true
for d in $V; do
  out=build/sbench; [ "$d" != 0 ] && out=build/sbench_d$d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -DTRX_DEV -DTRX_SB_DBG=$d "$@" -I include tools/sbench.hip -o $out 2>/dev/null &
done
wait
ls -la build/sbench*
