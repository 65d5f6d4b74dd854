#!/bin/bash
# same-box A/B of the FPN heads' RoIAlign form (roi_table 1 = default: roi_prep + table-driven channel-slice launch; 0: one workgroup per RoI):
# Mask R-CNN R50-FPN fp32 bs=2 and R101-FPN fp16 bs=8
mkdir -p gpurun_out/r5n
for rep in 1 2 3; do
  for g in 0 1 2 3; do
    timeout -k 10 200 python bench.py --model maskrcnn --steps 40 --warmup 8 --no-cpu-baseline --no-h2d --no-e2e --no-box --param roi_table=$g > gpurun_out/r5n/mrcnn_t${g}_$rep.json 2>> gpurun_out/r5n/err.txt || exit 1
    timeout -k 10 200 python bench.py --model maskrcnn --depth 101 --fp16 --batch 8 --steps 40 --warmup 8 --no-cpu-baseline --no-h2d --no-e2e --no-box --param roi_table=$g > gpurun_out/r5n/r101_t${g}_$rep.json 2>> gpurun_out/r5n/err.txt || exit 1
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5n/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], 'frac', d['roofline']['frac'], 'bs1', d.get('bs1', {}).get('p50_ms_per_image'))
    for h in d.get('roofline_hbm', []):
        if 'roi_align' in h['kernel']:
            print('    ', h['kernel'], h['us'], h['achieved'], h['frac'])
PY
