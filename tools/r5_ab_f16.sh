#!/bin/bash
# same-box A/B of the fp16 MFMA shape (R101 fp16 bs=8, configs[4] per-GPU shape) + the batched RPN selection (R50 fp32 bs=2 / bs=1)
mkdir -p gpurun_out/r5d
for rep in 1 2; do
  for shape in 0 1 2; do
    timeout -k 10 200 python bench.py --model maskrcnn --depth 101 --fp16 --batch 8 --steps 30 --warmup 8 --no-cpu-baseline --no-h2d --no-e2e --no-box --f16-mfma-shape $shape > gpurun_out/r5d/r101_shape${shape}_$rep.json 2>> gpurun_out/r5d/err.txt || exit 1
  done
done
for rep in 1 2; do
  for g in 0 -1; do
    timeout -k 10 200 python bench.py --model maskrcnn --steps 30 --warmup 8 --no-cpu-baseline --no-h2d --no-e2e --no-box --param rpn_select_groups=$g > gpurun_out/r5d/r50_groups${g}_$rep.json 2>> gpurun_out/r5d/err.txt || exit 1
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5d/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    rs = [h for h in d['roofline_hbm'] if h['kernel'].startswith('rpn_select')]
    print(f.split('/')[-1], d['value'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'], 'bs1', d['bs1']['p50_ms_per_image'], 'rpn_select us', rs[0]['us'] if rs else None, rs[0]['launches'] if rs else None)
PY
