# round 6: strictly alternating two-part phase C (ZKP_COOP_C_ALT existed only in the experiment build of this sweep: profiles/r06/knob_sweeps.txt r6i)
set -o pipefail
out=gpurun_out/r6i; mkdir -p $out; i=0
for cfg in "ZKP_NOP=1" "ZKP_COOP_C_ALT=1" "ZKP_COOP_C_ALT=1 ZKP_COOP_C_SPLIT_MIN=65536" "ZKP_COOP_C_ALT=1 ZKP_COOP_C_SPLIT_MIN=32768" "ZKP_COOP_C_ALT=0 ZKP_COOP_C_SPLIT_MIN=65536" "ZKP_NOP=2" "ZKP_COOP_C_ALT=1 ZKP_COOP_C_SPLIT_MIN=65536"; do
  i=$((i+1)); env $cfg python3 tools/batch_sweep.py --sizes 32768,65536,131072,262144,524288,1048576 --tag "$cfg" > $out/knobs_$i.json 2>> $out/knobs.err || exit 1
done
