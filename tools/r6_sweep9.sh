# round 6: which kernel classes profit from a primer (ZKP_COOP_PRIME=2 = every grid of 1 .. 12 workgroups per compute unit; mask: 1 k_coop, 2 k_prep_lines, 4 k_ksq, 8 k_batch_inv, 16 k_kdec_*)
set -o pipefail
out=gpurun_out/r6o; mkdir -p $out; i=0
for cfg in "ZKP_COOP_PRIME=0" "ZKP_COOP_PRIME=2 ZKP_COOP_PRIME_MASK=1" "ZKP_COOP_PRIME=2 ZKP_COOP_PRIME_MASK=2" "ZKP_COOP_PRIME=2 ZKP_COOP_PRIME_MASK=4" "ZKP_COOP_PRIME=2 ZKP_COOP_PRIME_MASK=8" "ZKP_COOP_PRIME=2 ZKP_COOP_PRIME_MASK=16" "ZKP_COOP_PRIME=2 ZKP_COOP_PRIME_MASK=31" "ZKP_COOP_PRIME=2 ZKP_COOP_PRIME_MASK=7" "ZKP_COOP_PRIME=0"; do
  i=$((i+1)); env $cfg python3 tools/batch_sweep.py --k 1 --sizes 1024,4096,8192,12288,16384,20480,24576,32768,40960,49152,65536,131072 --tag "$cfg" > $out/knobs_$i.json 2>> $out/knobs.err || exit 1
done
