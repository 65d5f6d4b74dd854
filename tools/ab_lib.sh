#!/bin/bash
# same-box A/B of two builds of libisegmi.so on the R101 fp16 bs 8 bench: tools/ab_lib.sh <other lib> <outdir>  (two runs each, alternating)
other=$1; out=${2:-gpurun_out/ab}; mkdir -p $out
for rep in 1 2; do
  ISEGMI_LIB=$other timeout -k 10 300 python bench.py --model maskrcnn --depth 101 --fp16 --batch 8 --no-cpu-baseline --no-latency --no-h2d --no-e2e > $out/other_$rep.json 2> $out/other_$rep.err || exit 1
  timeout -k 10 300 python bench.py --model maskrcnn --depth 101 --fp16 --batch 8 --no-cpu-baseline --no-latency --no-h2d --no-e2e > $out/this_$rep.json 2> $out/this_$rep.err || exit 1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["roofline"]["frac"], j["roofline"]["achieved"])
PY
