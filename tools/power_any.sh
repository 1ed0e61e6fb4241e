#!/bin/bash
# development: sample GPU power / clock (rocm-smi) while a command runs.  usage: tools/power_any.sh <command ...>
"$@" > /tmp/pa_out.txt 2>&1 &
pid=$!
sleep 2.0
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '
  echo
  sleep 0.4
done
wait $pid
tail -4 /tmp/pa_out.txt
