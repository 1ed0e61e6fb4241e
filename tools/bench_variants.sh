#!/bin/bash
# development: bench.py (no CPU baseline) alternately with the product library and several build/libtrx_<name>.so
#   bash tools/bench_variants.sh "zsnt zssc1" [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
names=$1; rounds=${2:-2}
cp $R/torchregister_amd/lib/libtrx.so /tmp/libtrx_orig.so
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'], 4), round(d['roofline']['kernel_ms'], 4), 'rot', round(d['config']['value_rot']), 'rigid', round(d['config']['value_rigid_randinit']), 'theta*', round(d['config']['value_theta_star']), 'run', round(d['config'].get('value_run') or 0), d['config'].get('body_histogram'))"; }
for i in $(seq $rounds); do
  for n in base $names; do
    if [ $n = base ]; then cp /tmp/libtrx_orig.so $R/torchregister_amd/lib/libtrx.so; else cp $R/build/libtrx_$n.so $R/torchregister_amd/lib/libtrx.so; fi
    python3 $R/bench.py --no-cpu-baseline 2>/dev/null | tail -1 | show $n
  done
done
cp /tmp/libtrx_orig.so $R/torchregister_amd/lib/libtrx.so
