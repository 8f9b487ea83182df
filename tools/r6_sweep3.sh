#!/bin/bash
# round 6, third GPU call: inversion lanes and an occupancy cap (LDS padding of k_coop) over small / medium batches
set -o pipefail
out=gpurun_out/r6d; mkdir -p $out
sizes=4096,8192,12288,16384,24576,32768,65536,131072,262144
i=0
for cfg in "ZKP_COOP_STAGGER=0" "ZKP_COOP_STAGGER=0 ZKP_COOP_INV_LANES=65536" "ZKP_COOP_STAGGER=0 ZKP_COOP_INV_LANES=131072" "ZKP_COOP_STAGGER=0 ZKP_COOP_INV_LANES=262144" \
           "ZKP_COOP_STAGGER=0 ZKP_COOP_INV_LANES=16384" "ZKP_COOP_STAGGER=0 ZKP_COOP_LDS_PAD=8192" "ZKP_COOP_STAGGER=0 ZKP_COOP_LDS_PAD=28000" "ZKP_COOP_STAGGER=0"; do
  i=$((i+1))
  env $cfg python3 tools/batch_sweep.py --sizes $sizes --tag "$cfg" > $out/knobs_$i.json 2>> $out/knobs.err || exit 1
  echo "knobs $i ($cfg) done"
done
