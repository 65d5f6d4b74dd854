#!/bin/bash
# Same-box A/B of the fp16 conv kernels: the round-2 build (instancesegmentation-jittor_amd/lib_r2/libisegmi.so, built from commit 7e54175
# by `git archive 7e54175 | make`) against the current one, on the R101 bs=8 layer shapes (tools/conv_f16_bench.py), with and without residual.
set -e
cd "$(dirname "$0")/.."
OLD=instancesegmentation-jittor_amd/lib_r2/libisegmi.so
for mode in "res 2,5,9,11" "nores 0,1,7,8,10,12"; do
  set -- $mode
  arg=$([ "$1" = res ] && echo res || echo x)
  echo "== $1: round-2 build"; ISEGMI_LIB=$OLD python tools/conv_f16_bench.py 0,37 $arg $2
  echo "== $1: current build"; python tools/conv_f16_bench.py 0,37 $arg $2
done
