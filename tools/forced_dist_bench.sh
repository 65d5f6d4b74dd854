#!/bin/bash
# bench.py through its N > 1 code path on ONE GPU (world-1 RCCL communicators, ISEGMI_BENCH_FORCE_DIST=1), next to the plain N = 1 run: same box
set -e -o pipefail
mkdir -p gpurun_out/fd
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-latency "$@" > gpurun_out/fd/plain.json 2> gpurun_out/fd/plain.err
ISEGMI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-latency "$@" > gpurun_out/fd/forced.json 2> gpurun_out/fd/forced.err
python - <<'PY'
import json
for n in ("plain", "forced"):
    d = json.loads([l for l in open("gpurun_out/fd/%s.json" % n) if l.startswith("{")][0])
    print("%-7s value %.1f resident %.1f e2e %.1f  rccl_ranks %s" % (n, d["value"], d.get("value_resident", 0), d.get("value_e2e", 0), d.get("rccl_ranks")))
PY
