set -e
for d in 1 2 3; do
  ISEGMI_BENCH_DEPTH=$d python bench.py --model maskrcnn --depth 101 --fp16 --batch 8 --steps 40 --warmup 10 --no-cpu-baseline --no-latency --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r101f16 depth $d value %.1f resident %.1f'%(d['value'],d.get('value_resident',0)))"
done
