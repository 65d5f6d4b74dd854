#!/bin/bash
# dev experiment: XCD band tile order (conv_set_band) against the round-2 rule (ISEGMI_CONV_BAND=0), same box: traffic and conv time
set -e -o pipefail
mkdir -p gpurun_out/band
for b in 0 1; do
  export ISEGMI_CONV_BAND=$b
  bash tools/conv_traffic.sh yolact_band$b yolact 8 > /dev/null
  bash tools/conv_traffic.sh maskrcnn_band$b maskrcnn 2 > /dev/null
  for rep in 1 2; do
    python tools/conv_report.py 8 0 yolact > gpurun_out/band/report_yolact8_band${b}_$rep.txt
    python tools/conv_report.py 2 0 maskrcnn > gpurun_out/band/report_maskrcnn2_band${b}_$rep.txt
    python tools/conv_report.py 1 0 yolact > gpurun_out/band/report_yolact1_band${b}_$rep.txt
    python tools/conv_report.py 1 0 maskrcnn > gpurun_out/band/report_maskrcnn1_band${b}_$rep.txt
  done
  echo "== band $b"; head -1 gpurun_out/ct/conv_traffic_yolact_band$b.txt; head -1 gpurun_out/ct/conv_traffic_maskrcnn_band$b.txt
  head -qn1 gpurun_out/band/report_*_band${b}_*.txt
done
