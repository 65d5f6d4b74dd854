#!/bin/bash
# Kernel-stats-only rocprofv3 pass of bench.py (single-stream), dev tool: tools/quick_stats.sh <tag> <bench args...>
set -e -o pipefail
tag=$1; shift
root=$(pwd); out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/qs_${tag} -o run -- python3 $root/bench.py $* --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-maskrcnn --no-h2d --single-stream > $out/qs_${tag}.log 2>&1
cd $root
rm -f $out/qs_${tag}/run_kernel_trace.csv
python3 - <<EOF
import csv
rows=list(csv.DictReader(open('$out/qs_${tag}/run_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('$tag total ms %.3f'%(tot/1e6))
for r in rows[:24]:
    print('  %-64s calls %5s avg %9.1f us  %5.2f%%'%(r['Name'][:64],r['Calls'],float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
EOF
