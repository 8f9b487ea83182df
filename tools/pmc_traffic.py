#!/usr/bin/env python3
"""HBM traffic of one bench.py pass from two rocprofv3 counter runs (FETCH_SIZE and WRITE_SIZE cannot share a
pass on gfx950: /opt/skills/guides/MI355X_MICROARCH.md, HBM + counter-slot table).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --bare
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --bare
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --pairs 1048576 > profiles/r02/pmc/traffic.json

Counters are in KB.  gfx950 correction: FETCH_SIZE counts the 128-B requests of 16 B/lane streaming reads at 64 B,
so it is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.  The profiled process runs several passes of
the pipeline (warm-up, timed steps, the HIP-event timing of the roofline leg); a pass is recognised by its single
k_batch_inv dispatch, and the input-generation kernels (k_g1_mul28 / k_g2_mul28) are left out."""
import argparse
import collections
import csv
import glob
import json
import os

PASS_KERNELS = ("k_prep_lines", "k_coop", "k_batch_inv", "k_ksq", "k_kdec_a", "k_kdec_b")


def short(name):
    for k in PASS_KERNELS + ("k_g1_mul28", "k_g2_mul28"):
        if k + "(" in name or k + "<" in name:      # k_coop is a template: k_coop<24, 34>(...)
            return k
    return name.split("(")[0]


def load(dirname, counter):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under " + dirname
    per_kernel = collections.Counter()
    dispatches = collections.Counter()
    seen = set()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            per_kernel[k] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dispatches[k] += 1
    return per_kernel, dispatches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--pairs", type=int, required=True, help="pairs per pass of the profiled run")
    ap.add_argument("--passes", type=int, default=0, help="passes in the profiled run (default: k_ksq dispatches / 5)")
    a = ap.parse_args()
    fetch, dfetch = load(a.fetch_dir, "FETCH_SIZE")
    write, dwrite = load(a.write_dir, "WRITE_SIZE")
    # five x-power chains per pass - per PART of phase C (two parts from 2^18 checks on since round 5: pass --passes explicitly then;
    # `bench.py --steps 1 --warmup 0 --bare` is exactly one pass)
    passes = a.passes or dfetch["k_ksq"] // 5
    assert passes and dfetch["k_ksq"] == dwrite["k_ksq"] and dfetch["k_ksq"] % (5 * passes) == 0, (dfetch, dwrite)
    per = {}
    total = 0.0
    for k in PASS_KERNELS:
        fx2 = 2.0 * fetch[k] * 1024 / passes
        w = write[k] * 1024 / passes
        per[k] = {"fetch_x2": fx2, "write": w, "dispatches_per_pass": dfetch[k] / passes}
        total += fx2 + w
    # the DESIGN's necessary bytes of a pass (NOT the algorithm's: that is 864 B of I/O per pairing): inputs 288 B + Gt 576 B + ok byte
    # per pair; line stream written and read once
    # (68 steps x 6 records x 64 B); per-check state records the kernels exchange: miller 12 stores; fexp_a 12 loads + 9 stores;
    # inversion 2; the phase C step programs' K_STATE loads and stores (counted from the generated plan); per x-power chain
    # k_ksq 8 loads + 6 x 8 stores, k_kdec_a 6 x (8 loads + 5 stores), k_batch_inv 6 x 3, k_kdec_b 6 x (13 loads + 4 stores)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    import coopgen
    prog_state = sum(len(st["lanes"]) for seg in coopgen.fexp_c_segments() for st in seg.steps
                     if st["op"] in (coopgen.OP_GLOAD, coopgen.OP_GSTORE) and st["kind"] == coopgen.K_STATE)
    n = a.pairs
    lines = 68 * 6 * 64 * 2
    # round 4 on: three snapshots per chain; round 5: k_kdec_a 8 loads + 3 stores (N twice, n), k_kdec_b 11 loads + 4 stores
    state = (12 + 12 + 9 + 2 + prog_state + 5 * (8 + 3 * 8 + 3 * 11 + 9 + 3 * 15)) * 64
    algo = n * (288 + 576 + 1 + lines + state)
    out = {
        "workload": "bench.py pass: %d pairs, cooperative family; passes in the profiled run: %d" % (n, passes),
        "source": "tools/pmc_traffic.py over rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs of `bench.py --steps 1 --warmup 0 --bare`)",
        "correction": "FETCH_SIZE (KB) doubled (gfx950 tallies 128-B requests of 16 B/lane streaming reads at 64 B); WRITE_SIZE (KB) as is",
        "pairs_per_step": n,
        "hbm_bytes_per_step": total,
        "per_kernel_bytes": per,
        "design_bytes_per_step": algo,
        "design_bytes_what": "what THIS design has to move: I/O + the line stream written and read once + the state records its kernels exchange "
                             "(renamed in round 6: rounds 2-5 called it algorithmic_bytes_per_step)",
        "algorithmic_io_bytes_per_step": n * 864,
        "traffic_over_algorithmic_io": total / (n * 864.0),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
