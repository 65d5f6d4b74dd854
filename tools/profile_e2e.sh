#!/bin/bash
# kernel stats of a bench run WITH the e2e region (RLE / record pack kernels included), single stream: gpurun_out/e2e/<tag>_kernel_stats.csv
set -e -o pipefail
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/e2e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o run -- python3 $root/bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-maskrcnn --no-h2d --single-stream > $out/$tag.log 2>&1
cd $root
cp $(find $out/prof_$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
rm -rf $out/prof_$tag
head -40 $out/${tag}_kernel_stats.csv
