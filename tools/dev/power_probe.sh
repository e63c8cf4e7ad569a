#!/bin/bash
# What does the chip draw, and at which shader clock, while a kernel loops?  Samples rocm-smi beside a command:
#   bash tools/dev/power_probe.sh <label> <command ...>
label=$1; shift
"$@" > /dev/null 2>&1 &
pid=$!
sleep 2.5
for k in 1 2 3 4; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|junction|Sensor edge" | sed "s/^/[$label] /" | tr -s ' ' | head -8
  sleep 0.7
done
kill $pid 2>/dev/null
wait $pid 2>/dev/null
