#!/bin/bash
# same-box A/B of the grouped conv launches (fp32): Yolact bs=8 (headline) + bs=1 latency, Mask R-CNN bs=2 + bs=1
mkdir -p gpurun_out/r5e
for rep in 1 2; do
  for g in 0 1; do
    timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-h2d --no-e2e --no-box --no-maskrcnn --param conv_groups=$g > gpurun_out/r5e/yolact_g${g}_$rep.json 2>> gpurun_out/r5e/err.txt || exit 1
    timeout -k 10 200 python bench.py --model maskrcnn --steps 30 --warmup 8 --no-cpu-baseline --no-h2d --no-e2e --no-box --param conv_groups=$g > gpurun_out/r5e/mrcnn_g${g}_$rep.json 2>> gpurun_out/r5e/err.txt || exit 1
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5e/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], 'frac', d['roofline']['frac'], 'conv_ms', d['roofline']['conv_ms_per_step'], 'launches', d['roofline']['launches_per_step'], 'bs1', d['bs1']['p50_ms_per_image'], d['bs1']['roofline']['frac'])
PY
