#!/bin/bash
# same-box A/B of the fused fp16 bottlenecks (R101 bs 8): fused_bottleneck 1 (all) / 2 (identity blocks only) / 0 (none), two runs each
mkdir -p gpurun_out/r4d
for rep in 1 2; do for m in 1 2 0; do
  timeout -k 10 300 python bench.py --model maskrcnn --depth 101 --fp16 --batch 8 --no-cpu-baseline --no-latency --no-h2d --no-e2e --param fused_bottleneck=$m > gpurun_out/r4d/ab_${m}_${rep}.json 2> gpurun_out/r4d/ab_${m}_${rep}.err || exit 1
done; done
