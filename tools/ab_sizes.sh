#!/bin/bash
# same-box A/B over batch sizes: tools/ab_sizes.sh "<pairs list>" spec [spec ...]   (spec = lib[:ENV=val[,ENV=val]])
sizes=$1; shift
for n in $sizes; do
  for rep in 1 2; do
    for spec in "$@"; do
      lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=$(echo ${spec#*:} | tr ',' ' ')
      v=$(env $envs ZKP_LIB_PATH=$PWD/$lib python bench.py --steps 4 --warmup 1 --bare --pairs $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms' % d['ms_per_step'])")
      echo "n=$n $spec  $v"
    done
  done
done
