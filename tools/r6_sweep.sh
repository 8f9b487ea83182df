#!/bin/bash
# round 6, first GPU call: where the small / medium batches stand (default knobs), what the existing knobs do to them, and the
# launch timeline of a single pairing.  Output: gpurun_out/r6a/
set -o pipefail
out=gpurun_out/r6a; mkdir -p $out
export TMPDIR=/tmp
python3 tools/batch_sweep.py --k 1,3 --tag default > $out/sweep_default.json 2> $out/sweep_default.err || exit 1
echo "default done"; tail -c 600 $out/sweep_default.json
sizes=4096,8192,16384,32768,65536,131072,262144
i=0
for cfg in "ZKP_NOP=1" "ZKP_COOP_CHUNK=32768" "ZKP_COOP_CHUNK=32768 ZKP_COOP_C_SPLIT=2" "ZKP_COOP_CHUNK=16384 ZKP_COOP_C_SPLIT=2" \
           "ZKP_COOP_CHUNK=16384" "ZKP_COOP_C_SPLIT=2" "ZKP_COOP_CHUNK=8192 ZKP_COOP_C_SPLIT=2" "ZKP_COOP_STREAMS=3 ZKP_COOP_CHUNK=32768 ZKP_COOP_C_SPLIT=3" \
           "ZKP_COOP_STREAMS=4 ZKP_COOP_CHUNK=16384 ZKP_COOP_C_SPLIT=4" "ZKP_COOP_CHUNK=32768 ZKP_COOP_C_SPLIT=4" "ZKP_NOP=2"; do
  i=$((i+1))
  env $cfg python3 tools/batch_sweep.py --sizes $sizes --tag "$cfg" > $out/knobs_$i.json 2>> $out/knobs.err || exit 1
  echo "knobs $i ($cfg) done"
done
root=$PWD
(cd /tmp && rocprofv3 --kernel-trace -d $root/$out/trace_n1 -o t -- python3 $root/tools/batch_sweep.py --sizes 1 --reps 3 > $root/$out/trace_n1.log 2>&1) || exit 1
python3 tools/trace_small.py $(find $out/trace_n1 -name "*_results.db" | head -n 1) > $out/trace_n1.txt || exit 1
rm -rf $out/trace_n1
(cd /tmp && rocprofv3 --kernel-trace -d $root/$out/trace_n4096 -o t -- python3 $root/tools/batch_sweep.py --sizes 4096 --reps 3 > $root/$out/trace_n4096.log 2>&1) || exit 1
python3 tools/trace_small.py $(find $out/trace_n4096 -name "*_results.db" | head -n 1) > $out/trace_n4096.txt || exit 1
rm -rf $out/trace_n4096
tail -3 $out/trace_n1.txt $out/trace_n4096.txt
