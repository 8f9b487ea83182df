#!/bin/bash
# same-box A/B of library builds on the secondary workloads: tools/ab_workloads.sh <set a|b|c> ab/base.so ab/new.so [...]
set=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    v=$(ZKP_LIB_PATH=$PWD/$lib python tools/prof_workloads.py --set $set --warmup 1 --reps 3 2>/dev/null | tail -1)
    echo "$lib  $v"
  done
done
