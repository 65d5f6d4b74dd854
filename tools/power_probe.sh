#!/bin/bash
# Dev tool: sample rocm-smi (power, sclk) while a kernel loops.  usage: tools/power_probe.sh <out> <python args...>
out=$1; shift
python "$@" > $out.run 2>&1 &
pid=$!
sleep 4
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i -E "power|sclk|junction|edge" >> $out
  echo "--" >> $out
  sleep 1
done
wait $pid
