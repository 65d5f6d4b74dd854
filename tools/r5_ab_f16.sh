#!/bin/bash
# same-box A/B of the fp16 tile / MFMA-shape settings on the R101 fp16 bs=8 step (configs[4] per-GPU shape): 0 = 32x32x16 everywhere, 1 = row strips on
# 16x16x32, 3 = 1 + the 144-row tiles (library default)
mkdir -p gpurun_out/r5d
rm -f gpurun_out/r5d/r101_*.json
for rep in 1 2; do
  for shape in 0 1 3; do
    timeout -k 10 200 python bench.py --model maskrcnn --depth 101 --fp16 --batch 8 --steps 30 --warmup 8 --no-cpu-baseline --no-h2d --no-e2e --no-box --f16-mfma-shape $shape > gpurun_out/r5d/r101_shape${shape}_$rep.json 2>> gpurun_out/r5d/err.txt || exit 1
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5d/r101_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'], 'bs1', d['bs1']['p50_ms_per_image'])
PY
