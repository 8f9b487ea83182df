#!/bin/bash
# same-box A/B with the clock probe: pass time, sustained clock, kernel_ms of each build (bench.py without the CPU / secondary legs)
for rep in 1 2; do
  for lib in "$@"; do
    ZKP_LIB_PATH=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$lib  ms_per_step %.2f  kernel_ms %.2f  sustained_clock_ghz %.3f' % (d['ms_per_step'], r['kernel_ms'], r['sustained_clock_ghz']))"
  done
done
