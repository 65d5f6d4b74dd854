#!/bin/bash
# Round-6 rocprofv3 evidence for the three benched configurations, summaries written straight into gpurun_out/r06p/ (copied to profiles/ afterwards):
#   kernel stats (single stream), PMC FETCH_SIZE / WRITE_SIZE (separate passes, x2 wide-read correction in the summaries), MfmaUtil.
set -e -o pipefail
cd "$(dirname "$0")/.."
root=$(pwd)
mkdir -p gpurun_out/r06p
run() {  # tag, bench args
  tag=$1; shift
  bash tools/profile_round.sh $tag "$@" > gpurun_out/r06p/profile_$tag.log 2>&1
  cp $(find gpurun_out/prof_${tag}_stats -name "*kernel_stats.csv" | head -1) gpurun_out/r06p/r06_${tag}_kernel_stats.csv
  python tools/kernel_table.py gpurun_out/prof_${tag}_stats gpurun_out/prof_${tag}_fetch gpurun_out/prof_${tag}_write gpurun_out/r06p/r06_${tag}_kernel_table.md > /dev/null
  python tools/pmc_summary.py $(find gpurun_out/prof_${tag}_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/prof_${tag}_write -name "*counter_collection.csv" | head -1) gpurun_out/r06p/r06_pmc_${tag}.json
  python tools/mfma_util.py gpurun_out/prof_${tag}_mfma gpurun_out/r06p/r06_mfma_util_${tag}.txt > /dev/null
  python tools/hbm_stage_traffic.py gpurun_out/prof_${tag}_fetch gpurun_out/prof_${tag}_write 22 gpurun_out/r06p/r06_hbm_stage_traffic_${tag}.json > /dev/null || true
  rm -rf gpurun_out/prof_${tag}_stats gpurun_out/prof_${tag}_fetch gpurun_out/prof_${tag}_write gpurun_out/prof_${tag}_mfma
  echo "profiled $tag"
}
run yolact --model yolact
run maskrcnn --model maskrcnn
run r101f16 --model maskrcnn --depth 101 --fp16 --batch 8
