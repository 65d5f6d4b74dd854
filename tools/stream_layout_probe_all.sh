#!/bin/bash
mkdir -p gpurun_out/r4e
timeout -k 10 500 python tools/stream_layout_probe.py 0,1,2,3,4 > gpurun_out/r4e/layout5.txt 2>&1 || exit 1
for k in 0 1 4; do timeout -k 10 200 python tools/second_engine_probe.py 0 foreign=$k >> gpurun_out/r4e/layout5_m.txt 2>&1 || exit 1; done
