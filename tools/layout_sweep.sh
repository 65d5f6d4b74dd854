#!/bin/bash
# dev: alternative role -> queue-class layouts (ISEGMI_STREAM_LAYOUT: main side0 side1 side2 tail heads hs0 hs1 hs2 copy)
mkdir -p gpurun_out/r4e
for L in 0321123012 0321123011 0321123013 0021123012 0321121312 0321123010 0121323012; do
  echo "layout $L" >> gpurun_out/r4e/sweep.txt
  ISEGMI_STREAM_LAYOUT=$L timeout -k 10 100 python tools/stream_layout_probe.py 0 >> gpurun_out/r4e/sweep.txt 2>&1 || exit 1
  ISEGMI_STREAM_LAYOUT=$L timeout -k 10 100 python tools/second_engine_probe.py 0 >> gpurun_out/r4e/sweep.txt 2>&1 || exit 1
done
