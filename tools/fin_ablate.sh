#!/bin/bash
# development: finalize-kernel ablations (build/libtrx_fin{1,2}.so swapped in for the product library on the GPU box's copy).
# Build them here first:  for a in 1 2; do make -B HIPFLAGS="<the Makefile's HIPFLAGS> -DTRX_FIN_ABLATE=$a"; cp torchregister_amd/lib/libtrx.so build/libtrx_fin$a.so; done; make -B
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/torchregister_amd/lib/libtrx.so /tmp/libtrx_orig.so
for a in 1 2; do
  cp $R/build/libtrx_fin$a.so $R/torchregister_amd/lib/libtrx.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin$a -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2>&1
  echo "== ablate $a"; grep finalize $R/gpurun_out/fin$a/*/*kernel_stats.csv | cut -d, -f2-8 | tail -2
done
cp /tmp/libtrx_orig.so $R/torchregister_amd/lib/libtrx.so
