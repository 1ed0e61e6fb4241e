#!/bin/bash
# development: bench.py (no CPU baseline) alternately with the product library and build/libtrx_<name>.so
#   bash tools/bench_variant.sh w6 [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
n=$1; rounds=${2:-2}
cp $R/torchregister_amd/lib/libtrx.so /tmp/libtrx_orig.so
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'], 4), round(d['roofline']['kernel_ms'], 4), 'rot', round(d['config']['value_rot']), 'rigid', round(d['config']['value_rigid_randinit']), 'flow', round(d['config']['flow_value']))"; }
for i in $(seq $rounds); do
  cp /tmp/libtrx_orig.so $R/torchregister_amd/lib/libtrx.so; python3 $R/bench.py --no-cpu-baseline 2>/dev/null | tail -1 | show base
  cp $R/build/libtrx_$n.so $R/torchregister_amd/lib/libtrx.so; python3 $R/bench.py --no-cpu-baseline 2>/dev/null | tail -1 | show $n
done
cp /tmp/libtrx_orig.so $R/torchregister_amd/lib/libtrx.so
