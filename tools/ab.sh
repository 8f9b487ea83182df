#!/bin/bash
# A/B timing of library builds on ONE box: tools/ab.sh ab/base.so ab/new.so [...]   (bench.py --bare, 2^20 pairs, alternating;
# a build name may carry environment settings: "ab/new.so:ZKP_COOP_INV_LANES=65536")
for rep in 1 2; do
  for spec in "$@"; do
    lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=${spec#*:}
    v=$(env $envs ZKP_LIB_PATH=$PWD/$lib python bench.py --steps 3 --warmup 1 --bare 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms' % d['ms_per_step'])")
    echo "$spec  $v"
  done
done
