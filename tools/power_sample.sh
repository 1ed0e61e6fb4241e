#!/bin/bash
# development: sample GPU power / clocks (rocm-smi) while a zbench variant loops.  usage: tools/power_sample.sh <binary> [reps]
R=${GRAFT_REPO_ROOT:-/root/repo}
bin=$1; reps=${2:-20000}
$R/build/$bin 8 256 0.004 $reps x > /tmp/ps_$bin.txt 2>&1 &
pid=$!
sleep 1.5
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|memory)" | tr '\n' ';' | sed 's/  */ /g'
  echo
  sleep 0.5
done
wait $pid
tail -1 /tmp/ps_$bin.txt
