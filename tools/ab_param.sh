#!/bin/bash
# same-box A/B of one engine parameter on the R101 fp16 bs 8 bench: tools/ab_param.sh <param> <value A> <value B> <outdir>  (two runs each, alternating)
par=$1; a=$2; b=$3; out=${4:-gpurun_out/abp}; mkdir -p $out
for rep in 1 2; do for v in $a $b; do
  timeout -k 10 300 python bench.py --model maskrcnn --depth 101 --fp16 --batch 8 --no-cpu-baseline --no-latency --no-h2d --no-e2e --param $par=$v > $out/${par}_${v}_$rep.json 2> $out/${par}_${v}_$rep.err || exit 1
done; done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/${par}_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["achieved"])
PY
