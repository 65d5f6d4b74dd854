set -e
for p in 0 1; do
  for cfg in "--model maskrcnn" "--model maskrcnn --depth 101 --fp16 --batch 8" "--model maskrcnn --fp16 --batch 2"; do
    python bench.py $cfg --param rpn_select_on_tail=$p --steps 40 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rpn_select_on_tail $p  %-50s value %.1f e2e %.1f bs1 p50 %.2f ms' % ('$cfg', d['value'], d.get('value_e2e', 0), d.get('bs1', {}).get('p50_ms_per_image', 0)))"
  done
done
