#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for pose in inv rz0.1; do
  echo "== ZB_POSE=$pose (the 64 x 32 tile does not fit: the loop is the flat tile's)"
  ZB_POSE=$pose $R/build/zbench 8 256 0.004 20000 x > /tmp/zsf_$pose.txt 2>&1 &
  pid=$!
  sleep 2.5
  for i in 1 2 3 4 5; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5
  done
  wait $pid
  grep -E "flat  " /tmp/zsf_$pose.txt | tail -2
done
echo "== identity-ish (eps 0.004): the 64 x 32 tile"
$R/build/zbench 8 256 0.004 20000 x > /tmp/zs64.txt 2>&1 &
pid=$!
sleep 2.5
for i in 1 2 3; do /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done
wait $pid
grep -E "zstream 64x32  " /tmp/zs64.txt | tail -1
