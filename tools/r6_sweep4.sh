#!/bin/bash
# round 6: the phase-C split threshold again, on the build with the faster inversion
set -o pipefail
out=gpurun_out/r6g; mkdir -p $out
sizes=131072,262144,524288,1048576
i=0
for cfg in "ZKP_NOP=1" "ZKP_COOP_C_SPLIT_MIN=131072" "ZKP_COOP_C_SPLIT_MIN=524288" "ZKP_COOP_C_SPLIT_MIN=100000000" "ZKP_NOP=2" "ZKP_COOP_C_SPLIT_MIN=131072"; do
  i=$((i+1))
  env $cfg python3 tools/batch_sweep.py --sizes $sizes --tag "$cfg" > $out/knobs_$i.json 2>> $out/knobs.err || exit 1
  echo "knobs $i ($cfg) done"
done
