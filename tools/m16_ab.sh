#!/bin/bash
# round 5: v_mfma_f32_16x16x32_f16 against v_mfma_f32_32x32x16_f16 -- microbenchmark, per-layer A/B of the tiles, then the R101 fp16 bs=8 step under both shapes
set -e
mkdir -p gpurun_out/r5b
(cd tools/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && timeout -k 10 120 ./mfma_shape) > gpurun_out/r5b/mfma_shape.txt 2>&1
timeout -k 10 300 python tools/conv_f16_bench.py 30,40 none 0,1,8 > gpurun_out/r5b/strip_ab.txt 2>&1
timeout -k 10 300 python tools/conv_f16_bench.py 34,44,37,47,39,49 res 2,3,5,7,9,11 > gpurun_out/r5b/persist_ab.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_conv_f16_gpu.py -x -q -m gpu -k "40 or 44 or 47 or 49 or 2092 or 2095 or 2097 or shape" > gpurun_out/r5b/pytest_m16.txt 2>&1 || true
tail -3 gpurun_out/r5b/pytest_m16.txt
