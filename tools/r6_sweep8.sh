# round 6: primer launches (ZKP_COOP_PRIME: 0 off, 1 the two bad bands, 2 every grid up to 12 workgroups per compute unit)
set -o pipefail
out=gpurun_out/r6n; mkdir -p $out; i=0
for cfg in "ZKP_COOP_PRIME=0" "ZKP_COOP_PRIME=1" "ZKP_COOP_PRIME=2" "ZKP_COOP_PRIME=3" "ZKP_COOP_PRIME=4" "ZKP_COOP_PRIME=0" "ZKP_COOP_PRIME=2" "ZKP_COOP_PRIME=3"; do
  i=$((i+1)); env $cfg python3 tools/batch_sweep.py --k 1 --sizes 1,1024,2048,3072,4096,5120,6144,8192,10240,12288,16384,20480,24576,32768,40960,49152,65536,131072,262144 --tag "$cfg" > $out/knobs_$i.json 2>> $out/knobs.err || exit 1
done
