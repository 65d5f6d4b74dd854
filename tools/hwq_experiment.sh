#!/bin/bash
# GPU_MAX_HW_QUEUES sweep (the runtime's in-order hardware queues per process; default 4)
set -e
for q in 2 3 4 5 6 8; do
  export GPU_MAX_HW_QUEUES=$q
  for cfg in "--no-maskrcnn" "--model maskrcnn" "--model maskrcnn --depth 101 --fp16 --batch 8"; do
    python bench.py $cfg --steps 40 --warmup 10 --no-cpu-baseline --no-h2d --no-latency --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GPU_MAX_HW_QUEUES $q  %-50s value %.1f' % ('$cfg', d['value']))"
  done
done
