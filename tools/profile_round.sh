#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> <bench args...>     e.g.  tools/profile_round.sh yolact --model yolact
# The stats pass runs with --single-stream so that per-kernel durations are not inflated by overlapping launches and can be
# compared with bench.py's own HIP-event conv timing.  Passes (each its own process, as gpurun / the MI355X guide require): kernel stats; PMC FETCH_SIZE; PMC WRITE_SIZE;
# PMC SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE.  Raw output under gpurun_out/prof_<tag>_*; summaries are then made
# with tools/kernel_table.py, tools/pmc_summary.py and tools/mfma_util.py and copied into profiles/.
set -e -o pipefail
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
args="$root/bench.py $* --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-maskrcnn --no-h2d --no-e2e --no-box"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_stats -o run -- python3 $args --single-stream > $out/prof_${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/prof_${tag}_fetch -o run -- python3 $args > $out/prof_${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/prof_${tag}_write -o run -- python3 $args > $out/prof_${tag}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/prof_${tag}_mfma -o run -- python3 $args > $out/prof_${tag}_mfma.log 2>&1
cd $root
# keep only the small CSVs (the merge back is capped at 64 MiB): drop per-dispatch traces except the counter files
find $out/prof_${tag}_fetch $out/prof_${tag}_write $out/prof_${tag}_mfma -name "*kernel_trace.csv" -delete
echo "profile $tag done"
