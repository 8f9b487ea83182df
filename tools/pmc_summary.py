#!/usr/bin/env python3
"""Per-kernel reduction of a rocprofv3 SQ-counter run of `bench.py --steps 1 --warmup 0 --bare` (one pass of 2^20 pairs):

    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY \\
              --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --bare
    python3 tools/pmc_summary.py gpurun_out/pmc_sq [more dirs: further counter sets of the same command] > profiles/r02/pmc/pmc_summary.json

k_coop is split by its template arguments: <30,4> is the Miller program, <36,24> the hard part's step programs behind the compressed
squaring runs, <24,34> runs fexp_a per 2^16-check chunk (small grids) and the remaining phase C step programs (large grids).  Derived per group:
VALU instructions per wave, SIMD-cycles per VALU instruction (SQ_WAVE_CYCLES / waves-per-SIMD is not available per
dispatch, so the figure is SQ_BUSY_CYCLES-free: wave-cycles / VALU instructions of the same waves)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def group(name, grid, big):
    m = re.search(r"k_coop<\s*(\d+)\s*,\s*(\d+)\s*>", name)
    if m:
        cfg = (int(m.group(1)), int(m.group(2)))         # the instantiation's template arguments (slots, constants), not a substring
        if cfg == (30, 4):
            return "k_coop<30,4> miller"
        if cfg == (36, 24):
            return "k_coop<36,24> hard-part step programs"
        if cfg == (24, 34):
            return "k_coop<24,34> phase-C step programs" if grid >= big else "k_coop<24,34> fexp_a"
        return "k_coop<%d,%d>" % cfg
    for k in ("k_prep_lines", "k_batch_inv", "k_ksq", "k_kdec_a", "k_kdec_b", "k_g1_mul28", "k_g2_mul28", "k_set_int"):
        if k in name:
            return k
    return None


def main():
    rows = []
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows += list(csv.DictReader(open(f)))
    assert rows, "no counter_collection.csv found"
    big = max(int(r["Grid_Size"]) for r in rows if "k_coop" in r["Kernel_Name"]) // 2
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in rows:
        g = group(r["Kernel_Name"], int(r["Grid_Size"]), big)
        if g is None or "mul28" in g or g == "k_set_int":
            continue
        acc[g][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[g].add(r["Dispatch_Id"])
        acc[g]["_vgpr"] = max(acc[g]["_vgpr"], float(r["VGPR_Count"]))
    out = {"source": "tools/pmc_summary.py over rocprofv3 --pmc runs of `bench.py --steps 1 --warmup 0 --bare` (one pass, 2^20 pairs)", "groups": {}}
    for g, c in sorted(acc.items()):
        e = {k: v for k, v in c.items() if not k.startswith("_")}
        e["dispatches"] = len(disp[g])
        e["vgprs"] = c["_vgpr"]
        w, valu, cyc = c.get("SQ_WAVES", 0), c.get("SQ_INSTS_VALU", 0), c.get("SQ_WAVE_CYCLES", 0)
        if w and valu:
            e["valu_insts_per_wave"] = valu / w
        if valu and cyc:
            e["wave_cycles_per_valu_inst"] = cyc / valu
        if cyc and c.get("SQ_WAIT_ANY"):
            e["wait_any_share_of_wave_cycles"] = c["SQ_WAIT_ANY"] / cyc
        if cyc and c.get("SQ_ACTIVE_INST_VALU"):
            e["valu_active_share_of_wave_cycles"] = c["SQ_ACTIVE_INST_VALU"] / cyc
        out["groups"][g] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
