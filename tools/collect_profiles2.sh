#!/bin/bash
# Profiles of the workloads behind the bench line's secondary_workloads (BASELINE configs 4 / 5) and of the multi_miller_loop() ABI,
# on the GPU box, from the repo root:
#     bash tools/collect_profiles2.sh v40 [sets, default "a b"]
# Per set: one rocprofv3 kernel trace, two SQ counter passes, FETCH_SIZE and WRITE_SIZE (each its own run; counters are never
# combined with a trace).  Raw output -> gpurun_out/prof2_<tag>/ (scratch); summaries -> gpurun_out/profiles_<tag>/workloads_<set>.json,
# copied to profiles/rNN/ by hand after a look.
set -e -o pipefail
tag=${1:-v40}; sets=${2:-"a b"}
root=$PWD
raw=$root/gpurun_out/prof2_$tag; out=$root/gpurun_out/profiles_$tag
rm -rf "$raw"; mkdir -p "$raw" "$out"
export TMPDIR=/tmp
cd /tmp
W="$root/tools/prof_workloads.py"
for s in $sets; do
    rocprofv3 --kernel-trace --stats -d "$raw/trace_$s" -o t$s -- python3 $W --set $s --warmup 1 --reps 1 > "$raw/trace_$s.log" 2>&1
    echo "set $s: kernel trace done"
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY \
        --output-format csv -d "$raw/sq1_$s" -- python3 $W --set $s > "$raw/sq1_$s.log" 2>&1
    rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY \
        --output-format csv -d "$raw/sq2_$s" -- python3 $W --set $s > "$raw/sq2_$s.log" 2>&1
    echo "set $s: SQ counters done"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$raw/fetch_$s" -- python3 $W --set $s > "$raw/fetch_$s.log" 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$raw/write_$s" -- python3 $W --set $s > "$raw/write_$s.log" 2>&1
    echo "set $s: traffic passes done"
    python3 $root/tools/pmc_workloads.py --trace $(find "$raw/trace_$s" -name "*_results.db" | head -n 1) --sq "$raw/sq1_$s" "$raw/sq2_$s" \
        --fetch "$raw/fetch_$s" --write "$raw/write_$s" --bench "$raw/trace_$s.log" > "$out/workloads_$s.json"
done
echo "workload profiles written to $out"
