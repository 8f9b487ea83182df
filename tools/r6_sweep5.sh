# round 6: inversion lanes / batch on the faster inversion (profiles/r06/knob_sweeps.txt r6h)
set -o pipefail
out=gpurun_out/r6h; mkdir -p $out; i=0
for cfg in "ZKP_NOP=1" "ZKP_COOP_INV_LANES=65536" "ZKP_COOP_INV_LANES=131072" "ZKP_COOP_INV_LANES=16384" "ZKP_COOP_INV_BATCH=8" "ZKP_NOP=2"; do
  i=$((i+1)); env $cfg python3 tools/batch_sweep.py --sizes 16384,65536,131072,262144,1048576 --tag "$cfg" > $out/knobs_$i.json 2>> $out/knobs.err || exit 1
done
