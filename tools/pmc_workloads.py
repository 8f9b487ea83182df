#!/usr/bin/env python3
"""Per-kernel reduction of rocprofv3 runs of tools/prof_workloads.py (BASELINE configs 4 / 5, the multi_miller_loop() ABI):

    python3 tools/pmc_workloads.py --trace <results.db> --sq <dir> [<dir> ...] --fetch <dir> --write <dir> --bench <json line file>

Every argument is optional; what is given is reduced into ONE JSON document per workload set:
  kernels.<name>.trace    calls, total / avg / min / max ms, grid, VGPRs (rocprofv3's figure = half the allocation), LDS, scratch
  kernels.<name>.sq       the SQ counters summed over the kernel's dispatches + derived figures (VALU instructions per wave,
                          wave-cycles per VALU instruction, share of wave-cycles waiting / with the VALU active)
  kernels.<name>.traffic  FETCH_SIZE x 2 (gfx950 correction, /opt/skills/guides/MI355X_MICROARCH.md) and WRITE_SIZE in bytes
  roofline.<workload>     algorithmic multiply-adds (bench.py's SURVEY 8d counts) / summed kernel time / 39.3 T MAC/s - the
                          secondary_workloads.*.frac of the bench line, recomputable from this file
Template instantiations are kept apart (k_prep_lines<true> / <false>, k_coop<30,4> = Miller programs, k_coop<24,34> split into
fexp_a (small grids) and the phase C step programs)."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kname(name, grid=None, big=None):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    s = m.group(1) if m else name[:40]
    if s == "k_prep_lines":
        return s + ("<true>" if "<true>" in name or "Lb1" in name else "<false>")
    if s == "k_coop":
        m2 = re.search(r"k_coop<(\d+), ?(\d+)>", name) or re.search(r"k_coopILi(\d+)ELi(\d+)", name)
        if m2 and m2.group(1) == "30":
            return "k_coop<30,4> miller"
        if grid is not None and big is not None:
            return "k_coop<24,34> phase C" if grid >= big else "k_coop<24,34> fexp_a"
        return "k_coop<24,34>"
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace")
    ap.add_argument("--sq", nargs="*", default=[])
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--bench", help="file holding the JSON line tools/prof_workloads.py printed in the TRACE run")
    a = ap.parse_args()
    out = {"source": "tools/pmc_workloads.py over rocprofv3 runs of tools/prof_workloads.py", "kernels": collections.OrderedDict()}
    K = out["kernels"]
    if a.trace:
        cur = sqlite3.connect(a.trace).cursor()
        rows = list(cur.execute("select name, start, end, grid_x, vgpr_count, lds_size, scratch_size from kernels order by start"))
        coop = [r[3] for r in rows if "k_coop" in r[0] and "30" not in r[0].split("k_coop", 1)[1][:12]]
        big = max(coop) // 2 if coop else None
        for r in rows:
            k = kname(r[0], r[3], big)
            e = K.setdefault(k, {}).setdefault("trace", {"calls": 0, "total_ms": 0.0, "min_ms": 1e30, "max_ms": 0.0, "grid_x": r[3],
                                                         "vgpr_rocprof": r[4], "lds_bytes": r[5], "scratch": r[6]})
            d = (r[2] - r[1]) / 1e6
            e["calls"] += 1
            e["total_ms"] += d
            e["min_ms"] = min(e["min_ms"], d)
            e["max_ms"] = max(e["max_ms"], d)
            e["grid_x"] = max(e["grid_x"], r[3])
        for k in K:
            t = K[k]["trace"]
            t["avg_ms"] = t["total_ms"] / t["calls"]

    def counter_rows(d):
        rows = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows += list(csv.DictReader(open(f)))
        return rows

    sq_rows = []
    for d in a.sq:
        sq_rows += counter_rows(d)
    if sq_rows:
        coop = [int(r["Grid_Size"]) for r in sq_rows if "k_coop" in r["Kernel_Name"] and "30" not in r["Kernel_Name"].split("k_coop", 1)[1][:12]]
        big = max(coop) // 2 if coop else None     # Grid_Size is in work-items here; only the ratio matters
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in sq_rows:
            k = kname(r["Kernel_Name"], int(r["Grid_Size"]), big)
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            acc[k]["_vgpr"] = max(acc[k]["_vgpr"], float(r["VGPR_Count"]))
        for k, c in acc.items():
            e = {n: v for n, v in c.items() if not n.startswith("_")}
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
            K.setdefault(k, {})["sq"] = e
    for (d, cname, key, mult) in ((a.fetch, "FETCH_SIZE", "fetch_x2_bytes", 2.0), (a.write, "WRITE_SIZE", "write_bytes", 1.0)):
        if not d:
            continue
        rows = counter_rows(d)
        coop = [int(r["Grid_Size"]) for r in rows if "k_coop" in r["Kernel_Name"] and "30" not in r["Kernel_Name"].split("k_coop", 1)[1][:12]]
        big = max(coop) // 2 if coop else None
        per = collections.Counter()
        for r in rows:
            if r["Counter_Name"] == cname:
                per[kname(r["Kernel_Name"], int(r["Grid_Size"]), big)] += float(r["Counter_Value"]) * 1024 * mult
        for k, v in per.items():
            K.setdefault(k, {}).setdefault("traffic", {})[key] = v
    if a.bench and a.trace:
        import bench
        with open(a.bench) as f:
            line = json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])
        out["run"] = line
        reps, n = line["launches"], line["n"]
        roof = out["roofline"] = {"peak_T_mac_per_s": bench.PEAK_MACS / 1e12, "macs_per_fp_mul_equivalent": bench.MACS_PER_FPMUL,
                                  "note": "frac = items x Fp-mul-equivalents x 300 / (summed kernel time of ONE launch of the workload) / peak; "
                                          "the input generation's k_g1_mul28 / k_g2_mul28 dispatches are counted with their own items"}

        def ms(keys, per_launch_calls=None):
            t = 0.0
            for k in keys:
                if k in K and "trace" in K[k]:
                    t += K[k]["trace"]["total_ms"]
            return t / reps

        if line["set"] == "a":
            nc = line["config4_checks"]
            c4 = ["k_prep_lines<true>", "k_coop<30,4> miller", "k_coop<24,34> fexp_a", "k_coop<24,34> phase C", "k_batch_inv", "k_ksq", "k_kdec_a",
                  "k_kdec_b", "k_set_int"]
            t4 = ms(c4)
            roof["config4_three_pair_checks"] = {"checks": nc, "fp_mul_equivalents_per_check": bench.FPMUL_CHECK3, "kernel_ms_sum": t4,
                                                 "wall_ms": line["config4_ms"],
                                                 "frac_kernel_sum": nc * bench.FPMUL_CHECK3 * bench.MACS_PER_FPMUL / (t4 * 1e-3) / bench.PEAK_MACS,
                                                 "frac_wall": nc * bench.FPMUL_CHECK3 * bench.MACS_PER_FPMUL / (line["config4_ms"] * 1e-3) / bench.PEAK_MACS,
                                                 "note": "the two pipelines overlap kernels, so the SUM of kernel times exceeds the wall time"}
            for (nm, ks, fpm) in (("config5_g1_is_valid", ["k_g1_valid_fast", "k_g1_valid28"], bench.FPMUL_G1_VALID),
                                  ("config5_g2_is_valid", ["k_g2_valid_fast3", "k_g2_valid_fast", "k_g2_valid28"], bench.FPMUL_G2_VALID)):
                t = ms(ks)       # the asm kernel + the generic kernel's pass over the points it marked
                roof[nm] = {"points": n, "fp_mul_equivalents_per_point": fpm, "kernel_ms": t, "kernel": " + ".join(ks),
                            "frac": n * fpm * bench.MACS_PER_FPMUL / (t * 1e-3) / bench.PEAK_MACS}
        elif line["set"] == "b":
            t = ms(["k_prep_lines<false>", "k_coop<30,4> miller"])
            roof["multi_miller_loop"] = {"pairs": n, "fp_mul_equivalents_per_pair": bench.FPMUL_MILLER, "kernel_ms_sum": t, "wall_ms": line["multi_miller_loop_ms"],
                                         "frac_wall": n * bench.FPMUL_MILLER * bench.MACS_PER_FPMUL / (line["multi_miller_loop_ms"] * 1e-3) / bench.PEAK_MACS}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
