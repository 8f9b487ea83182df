# round 6: the shipped primer rule (ZKP_COOP_PRIME=1) against none, k = 1 and 3
set -o pipefail
out=gpurun_out/r6p; mkdir -p $out; i=0
for cfg in "ZKP_COOP_PRIME=0" "ZKP_COOP_PRIME=1" "ZKP_COOP_PRIME=0" "ZKP_COOP_PRIME=1"; do
  i=$((i+1)); env $cfg python3 tools/batch_sweep.py --k 1,3 --sizes 1,4096,8192,10240,12288,14336,16384,20480,24576,28672,32768,40960,49152,65536 --tag "$cfg" > $out/knobs_$i.json 2>> $out/knobs.err || exit 1
done
