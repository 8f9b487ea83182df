#!/bin/bash
# Collects everything profiles/rNN/ holds for the bench workload, on the GPU box, from the repo root:
#     bash tools/collect_profiles.sh r02 v22
# (one rocprofv3 kernel trace, four counter passes - each its own run, never combined with a trace - the step timings,
# the bench line).  Raw profiler output goes to gpurun_out/prof_<tag>/ (scratch); the summaries are written to
# gpurun_out/profiles_<tag>/ and copied to profiles/<round>/ by hand after a look.
set -e -o pipefail
round=${1:-r06}; tag=${2:-v60}
root=$PWD
raw=$root/gpurun_out/prof_$tag; out=$root/gpurun_out/profiles_$tag
rm -rf "$raw" "$out"; mkdir -p "$raw" "$out/pmc"
export TMPDIR=/tmp
cd /tmp
B="$root/bench.py"
rocprofv3 --kernel-trace --stats -d "$raw/trace" -o $tag -- python3 $B --steps 2 --warmup 1 --bare > "$raw/trace.log" 2>&1
python3 $root/tools/rocpd_stats.py $(find "$raw/trace" -name "*_results.db" | head -n 1) > "$out/${tag}_kernel_stats.txt"
echo "kernel trace done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY \
    --output-format csv -d "$raw/pmc_sq1" -- python3 $B --steps 1 --warmup 0 --bare > "$raw/pmc_sq1.log" 2>&1
echo "pmc set 1 done"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY \
    --output-format csv -d "$raw/pmc_sq2" -- python3 $B --steps 1 --warmup 0 --bare > "$raw/pmc_sq2.log" 2>&1
echo "pmc set 2 done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$raw/pmc_fetch" -- python3 $B --steps 1 --warmup 0 --bare > "$raw/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$raw/pmc_write" -- python3 $B --steps 1 --warmup 0 --bare > "$raw/pmc_write.log" 2>&1
echo "traffic passes done"
cd $root
python3 tools/pmc_summary.py "$raw/pmc_sq1" "$raw/pmc_sq2" > "$out/pmc/pmc_summary.json"
python3 tools/pmc_traffic.py "$raw/pmc_fetch" "$raw/pmc_write" --pairs 1048576 --passes 1 > "$out/pmc/traffic.json"
for s in sq1:sq_set1 sq2:sq_set2 fetch:fetch_size write:write_size; do
    f=$(find "$raw/pmc_${s%%:*}" -name "*counter_collection.csv" | head -n 1)
    # the per-dispatch rows of the pass's kernels only (the input generation kernels are left out)
    grep -v "mul28" "$f" > "$out/pmc/${s##*:}_counter_collection.csv"
done
python3 tools/time_steps.py > "$out/${tag}_time_steps.txt" 2>&1
python3 bench.py > "$out/${tag}_bench.json" 2> "$raw/bench.err"
echo "profiles for $round written to $out"
