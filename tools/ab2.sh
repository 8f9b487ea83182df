#!/bin/bash
# per-kernel / per-phase A/B of library builds on ONE box: tools/ab2.sh ab/base.so ab/new.so [...]   (tools/split.py per build, twice)
for rep in 1 2; do
  for lib in "$@"; do
    echo "$lib $(ZKP_LIB_PATH=$PWD/$lib python tools/split.py 2>/dev/null)"
  done
done
