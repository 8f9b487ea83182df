#!/bin/bash
# round 6, second GPU call: the stagger knobs (ZKP_COOP_STAGGER, ZKP_COOP_C_SPLIT_MIN, ZKP_COOP_SPLIT_MIN) over shard sizes
# NOTE: ZKP_COOP_STAGGER existed only in the experiment build of this sweep (removed: profiles/r06/knob_sweeps.txt r6b)
set -o pipefail
out=gpurun_out/r6b; mkdir -p $out
sizes=16384,32768,65536,131072,262144,524288,1048576
i=0
for cfg in "ZKP_COOP_STAGGER=0 ZKP_COOP_SPLIT_MIN=100000000" "ZKP_COOP_STAGGER=0" "ZKP_COOP_STAGGER=1" "ZKP_COOP_STAGGER=2" \
           "ZKP_COOP_STAGGER=0 ZKP_COOP_C_SPLIT_MIN=32768" "ZKP_COOP_STAGGER=1 ZKP_COOP_C_SPLIT_MIN=32768" "ZKP_COOP_STAGGER=2 ZKP_COOP_C_SPLIT_MIN=32768" \
           "ZKP_COOP_STAGGER=1 ZKP_COOP_C_SPLIT_MIN=32768 ZKP_COOP_SPLIT_MIN=100000000" \
           "ZKP_COOP_STAGGER=2 ZKP_COOP_C_SPLIT_MIN=32768 ZKP_COOP_SPLIT_MIN=8192" "ZKP_COOP_STAGGER=0 ZKP_COOP_SPLIT_MIN=100000000"; do
  i=$((i+1))
  env $cfg python3 tools/batch_sweep.py --sizes $sizes --tag "$cfg" > $out/knobs_$i.json 2>> $out/knobs.err || exit 1
  echo "knobs $i ($cfg) done"
done
