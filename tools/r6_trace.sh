#!/bin/bash
# kernel-trace timelines of one pairing call at the sizes given: tools/r6_trace.sh outdir "ENV=.. ENV=.." n [n ...]
set -o pipefail
out=$1; cfg=$2; shift 2
mkdir -p $out; root=$PWD; export TMPDIR=/tmp
for n in "$@"; do
  (cd /tmp && env $cfg rocprofv3 --kernel-trace -d $root/$out/tr_$n -o t -- python3 $root/tools/batch_sweep.py --sizes $n --reps 3 > $root/$out/tr_$n.log 2>&1) || exit 1
  python3 tools/trace_small.py $(find $out/tr_$n -name "*_results.db" | head -n 1) > $out/trace_$n.txt || exit 1
  rm -rf $out/tr_$n
  tail -n 1 $out/trace_$n.txt
done
