#!/bin/bash
set -e -o pipefail
root=$(pwd); out=$root/gpurun_out/e2e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o run -- python3 $root/tools/e2e_host_time.py "$@" > $out/trace.log 2>&1
cd $root
python tools/e2e_timeline.py $(find $out/trace -name "*kernel_trace.csv" | head -1) > $out/timeline.txt
rm -rf $out/trace
head -5 $out/timeline.txt
