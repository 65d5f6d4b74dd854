#!/bin/bash
# 144-row fp16 tiles: parity, per-layer A/B at M = 33 600, then the R101 fp16 bs=8 step (same box, interleaved)
mkdir -p gpurun_out/r5g
timeout -k 10 600 python -m pytest tests/test_conv_f16_gpu.py -x -q -m gpu -k "41 or 46 or 2094 or shape" > gpurun_out/r5g/pytest.txt 2>&1; tail -3 gpurun_out/r5g/pytest.txt
grep -q failed gpurun_out/r5g/pytest.txt && exit 1
timeout -k 10 200 python tools/conv_f16_bench.py 30,40,41 none 8 > gpurun_out/r5g/strip.txt 2>&1
timeout -k 10 200 python tools/conv_f16_bench.py 37,47,46,0 none 7 >> gpurun_out/r5g/strip.txt 2>&1
cat gpurun_out/r5g/strip.txt
for rep in 1 2; do
  timeout -k 10 200 python bench.py --model maskrcnn --depth 101 --fp16 --batch 8 --steps 30 --warmup 8 --no-cpu-baseline --no-h2d --no-e2e --no-box > gpurun_out/r5g/r101_$rep.json 2>> gpurun_out/r5g/err.txt || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5g/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'], 'bs1', d['bs1']['p50_ms_per_image'])
PY
