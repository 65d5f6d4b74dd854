#!/bin/bash
# Same-box A/B of two builds of libisegmi.so through bench.py: tools/ab_lib.sh <other/libisegmi.so> <bench args...>
set -e
other=$1; shift
for rep in 1 2; do
  for lib in "" "$other"; do
    ISEGMI_LIB=$lib python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-latency --no-h2d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s value %.1f e2e %.1f conv ms/step %.3f frac %.4f' % ('${lib:-current build}', d['value'], d.get('value_e2e', 0), d['roofline']['conv_ms_per_step'], d['roofline']['frac']))"
  done
done
