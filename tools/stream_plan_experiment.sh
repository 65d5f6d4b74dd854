#!/bin/bash
# A/B of the heads stream group's side streams (engine parameter heads_side_streams: 1 = the three streams of rounds 1-2 always, 0 = never,
# -1 = default: only for batches of at most two images), same box.  Round 3's wider sweep (heads sides shared with the main group's / no side
# streams at all / one shared side stream / group heads created first) is recorded in profiles/r03_experiments.txt.
set -e
for p in 1 0 -1; do
  for cfg in "--no-maskrcnn" "--yolact-config base --no-maskrcnn" "--fp16 --no-maskrcnn"; do
    python bench.py $cfg --param heads_side_streams=$p --steps 40 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('heads_side_streams %2s  %-40s value %.1f e2e %.1f bs1 p50 %.2f ms' % ('$p', '$cfg', d['value'], d.get('value_e2e', 0), d.get('bs1', {}).get('p50_ms_per_image', 0)))"
  done
done
