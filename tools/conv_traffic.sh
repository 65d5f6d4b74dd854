#!/bin/bash
# Per-layer HBM traffic of the conv launches (see tools/conv_traffic.py); run through gpurun from the repo root:
#   tools/conv_traffic.sh <tag> <model> <batch> [fp16] [101]      -> gpurun_out/ct/conv_traffic_<tag>.txt
set -e -o pipefail
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/ct
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_fetch -o run -- python3 $root/tools/conv_traffic.py run "$@" > $out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_write -o run -- python3 $root/tools/conv_traffic.py run "$@" > $out/${tag}_write.log 2>&1
cd $root
python tools/conv_traffic.py join $out/${tag}_fetch.log $out/${tag}_fetch $out/${tag}_write "$@" > $out/conv_traffic_${tag}.txt
rm -rf $out/${tag}_fetch $out/${tag}_write
head -3 $out/conv_traffic_${tag}.txt
