#!/bin/bash
# two SQ counter passes over one bare bench pass for the library ZKP_LIB_PATH selects -> gpurun_out/pmcq_<tag>.json
tag=${1:-x}
root=$PWD
export TMPDIR=/tmp
raw=$root/gpurun_out/pmcq_raw_$tag
rm -rf $raw; mkdir -p $raw
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY \
    --output-format csv -d $raw/sq1 -- python3 $root/bench.py --steps 1 --warmup 0 --bare > $raw/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY \
    --output-format csv -d $raw/sq2 -- python3 $root/bench.py --steps 1 --warmup 0 --bare > $raw/sq2.log 2>&1
cd $root
python3 tools/pmc_summary.py $raw/sq1 $raw/sq2 > gpurun_out/pmcq_$tag.json
rm -rf $raw
