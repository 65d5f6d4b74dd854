#!/bin/bash
# Reproduce every number of DESIGN.md's status table on one MI355X (run through gpurun from the repo root):
#   tools/run_all_benches.sh            -> gpurun_out/all_<name>.json, one JSON line each
# configs[1] (headline), configs[2], R101 fp32, configs[4] shapes (fp16), the C4 yaml of README.md:263-273, the optional fp16 Yolact.
set -o pipefail
out=gpurun_out
mkdir -p $out
run() { name=$1; shift; python bench.py "$@" > $out/all_$name.json 2> $out/all_$name.err; python - "$out/all_$name.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("%-28s value %8.1f img/s (resident %8.1f, e2e %8.1f)  %7.3f ms/step  conv %6.1f TF/s (%.3f of peak)  bs1 %.2f ms  rle==host %s" % (
      sys.argv[1].split("all_")[1][:-5], d["value"], d.get("value_resident", float("nan")), d.get("value_e2e", float("nan")), d["ms_per_step"],
      d["roofline"]["achieved"], d["roofline"]["frac"], d.get("bs1", {}).get("p50_ms_per_image", float("nan")), d.get("e2e", {}).get("device_rle_equals_host_encoder")))
PY
}
run yolact_bs8
run maskrcnn_r50_bs2 --model maskrcnn
run maskrcnn_r101_bs2 --model maskrcnn --depth 101 --no-cpu-baseline
run maskrcnn_r50_fp16_bs2 --model maskrcnn --fp16 --batch 2
run maskrcnn_r101_fp16_bs8 --model maskrcnn --depth 101 --fp16 --batch 8
run maskrcnn_r50_c4_bs2 --model maskrcnn --c4 --steps 10
run yolact_fp16_bs8 --fp16 --no-cpu-baseline
run yolact_base_bs8 --yolact-config base --no-cpu-baseline
run yolact_im700_bs8 --yolact-config im700 --no-cpu-baseline
run yolact_darknet53_bs8 --yolact-config darknet53 --no-cpu-baseline
run yolact_plus_resnet50_bs8 --yolact-config plus_resnet50 --no-cpu-baseline
run yolact_plus_base_bs8 --yolact-config plus_base --no-cpu-baseline
