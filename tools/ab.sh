#!/bin/bash
# A/B timing of library builds on ONE box: tools/ab.sh ab/base.so ab/new.so [...]   (bench.py, 2^20 pairs, alternating)
for rep in 1 2; do
  for lib in "$@"; do
    v=$(ZKP_LIB_PATH=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms  %s' % (d['ms_per_step'], d['config']['gt_sample_bit_exact']))")
    echo "$lib  $v"
  done
done
