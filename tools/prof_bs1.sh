#!/bin/bash
# kernel trace of the bs = 1 forwards, single stream: per-kernel averages
set -e -o pipefail
root=$(pwd); out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in ${MODELS:-maskrcnn yolact}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bs1_$m -o run -- python3 $root/bench.py --model $m --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-maskrcnn --no-h2d --no-e2e --no-box --single-stream > $out/prof_bs1_$m.log 2>&1
  f=$(find $out/prof_bs1_$m -name "*kernel_stats.csv" | head -1)
  cp $f $out/r6_bs1_${m}_kernel_stats.csv
  rm -rf $out/prof_bs1_$m
done
cd $root
head -45 gpurun_out/r6_bs1_maskrcnn_kernel_stats.csv | cut -c1-160
