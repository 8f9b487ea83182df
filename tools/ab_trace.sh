#!/bin/bash
# rocprofv3 kernel-trace A/B of library builds on ONE box: tools/ab_trace.sh ab/base.so ab/new.so   -> gpurun_out/abtrace_<name>.txt
root=$PWD
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  rm -rf $root/gpurun_out/abtrace_$name
  (cd /tmp && ZKP_LIB_PATH=$root/$lib rocprofv3 --kernel-trace --stats -d $root/gpurun_out/abtrace_$name -o t -- python3 $root/bench.py --steps 2 --warmup 1 --bare > $root/gpurun_out/abtrace_$name.log 2>&1)
  python3 tools/rocpd_stats.py $(find gpurun_out/abtrace_$name -name "*_results.db" | head -n 1) > gpurun_out/abtrace_$name.txt
  rm -rf $root/gpurun_out/abtrace_$name
  head -14 gpurun_out/abtrace_$name.txt
done
