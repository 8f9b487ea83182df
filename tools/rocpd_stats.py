#!/usr/bin/env python3
"""Kernel statistics and the timeline of the LAST pass from a rocprofv3 --kernel-trace run (rocpd SQLite output):
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o NAME -- python3 bench.py --steps 2 --warmup 1 --bare
    python3 tools/rocpd_stats.py gpurun_out/prof/NAME_results.db > profiles/rNN/NAME_kernel_stats.txt"""
import re
import sqlite3
import sys


def short(n):
    m = re.search(r"(k_[a-z0-9_]+)", n)
    s = m.group(1) if m else n[:48]
    m2 = re.search(r"k_coop<(\d+), ?(\d+)>", n) or re.search(r"k_coopILi(\d+)ELi(\d+)", n)
    return s + ("<%s,%s>" % (m2.group(1), m2.group(2)) if m2 else "")


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    rows = list(cur.execute("select name, start, end, grid_x, vgpr_count, lds_size, scratch_size from kernels order by start"))
    agg = {}
    for r in rows:
        a = agg.setdefault(short(r[0]), [0, 0.0, 1e30, 0.0, r[4], r[5], r[6]])
        d = (r[2] - r[1]) / 1e6
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    print("# whole run (all passes)\n%-28s %6s %12s %10s %10s %10s %6s %8s %8s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "vgpr", "lds_B", "scratch"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-28s %6d %12.3f %10.3f %10.3f %10.3f %6s %8s %8s" % (k, a[0], a[1], a[1] / a[0], a[2], a[3], a[4], a[5], a[6]))
    names = [short(r[0]) for r in rows]
    preps = [i for i, n in enumerate(names) if n == "k_prep_lines"]
    per_pass = 16
    seg = rows[preps[-per_pass]:]
    t0 = seg[0][1]
    print("\n# last pass: start_ms duration_ms kernel grid")
    for r in seg:
        print("%9.3f %9.3f %-28s %d" % ((r[1] - t0) / 1e6, (r[2] - r[1]) / 1e6, short(r[0]), r[3]))
    print("# pass wall %.3f ms" % ((max(r[2] for r in seg) - t0) / 1e6))


if __name__ == "__main__":
    main()
