#!/bin/bash
# SQ counters of one bare pass per library build, per kernel: tools/ab_pmc.sh ab/a.so ab/b.so -> gpurun_out/abpmc_<name>.json
root=$PWD
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  raw=$root/gpurun_out/abpmc_raw_$name
  rm -rf $raw; mkdir -p $raw
  (cd /tmp && ZKP_LIB_PATH=$root/$lib rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY \
      --output-format csv -d $raw/sq1 -- python3 $root/bench.py --steps 1 --warmup 0 --bare > $raw/sq1.log 2>&1)
  python3 tools/pmc_summary.py $raw/sq1 > gpurun_out/abpmc_$name.json
  rm -rf $raw
  python3 - <<PY
import json
d = json.load(open("gpurun_out/abpmc_$name.json"))["groups"]
for k in ("k_ksq", "k_coop<24,34> fexp_c step programs", "k_coop<30,4> miller", "k_kdec_a", "k_kdec_b"):
    g = d[k]
    print("$name %-36s waves %.0f wave_cycles %.4g busy_cycles %.4g valu %.4g wait_any %.3f" % (k, g["SQ_WAVES"], g["SQ_WAVE_CYCLES"], g["SQ_BUSY_CYCLES"], g["SQ_INSTS_VALU"], g.get("wait_any_share_of_wave_cycles", 0)))
PY
done
