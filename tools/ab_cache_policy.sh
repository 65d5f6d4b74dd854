#!/bin/bash
# Same-box A/B of the fp16 conv kernels' cache-policy constants (csrc/conv_mfma_f16.hip: CONV_F16_A_AUX / _RES_AUX / _OUT_AUX) through bench.py.
# Build the variants first (in the build container; the .so files travel with the snapshot):
#   tools/build_variant.sh a_nt -DCONV_F16_A_AUX=2 ; tools/build_variant.sh res_default -DCONV_F16_RES_AUX=0 ; tools/build_variant.sh out_nt -DCONV_F16_OUT_AUX=2
set -e
for v in "$@"; do echo "== $v"; bash tools/ab_lib.sh instancesegmentation-jittor_amd/lib_$v/libisegmi.so --model maskrcnn --depth 101 --fp16 --batch 8; done
