# round 6: small launches spread over the compute units by LDS padding (ZKP_COOP_SPREAD, default on) against the dispatcher's own placement
set -o pipefail
out=gpurun_out/r6l; mkdir -p $out; i=0
bash tools/r6_ksq_probe.sh > $out/ksq_probe_spread1.txt 2>&1 || exit 1
ZKP_COOP_SPREAD=0 bash tools/r6_ksq_probe.sh > $out/ksq_probe_spread0.txt 2>&1 || exit 1
for cfg in "ZKP_COOP_SPREAD=0" "ZKP_COOP_SPREAD=1" "ZKP_COOP_SPREAD=0" "ZKP_COOP_SPREAD=1"; do
  i=$((i+1)); env $cfg python3 tools/batch_sweep.py --sizes 1,1024,4096,8192,12288,16384,24576,32768,49152,65536,131072,262144,1048576 --tag "$cfg" > $out/knobs_$i.json 2>> $out/knobs.err || exit 1
done
