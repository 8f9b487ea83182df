#!/usr/bin/env python3
"""Timeline of the LAST pairing call in a rocprofv3 --kernel-trace run of tools/batch_sweep.py at a small n (rocpd SQLite output):
every launch with its start, duration and the idle gap in front of it - what a small batch pays in launch gaps against kernel time.
    python3 tools/trace_small.py <..._results.db>"""
import re
import sqlite3
import sys


def short(n):
    m = re.search(r"(k_[a-z0-9_]+)", n)
    s = m.group(1) if m else n[:40]
    m2 = re.search(r"k_coop<(\d+), ?(\d+)>", n) or re.search(r"k_coopILi(\d+)ELi(\d+)", n)
    return s + ("<%s,%s>" % (m2.group(1), m2.group(2)) if m2 else "")


cur = sqlite3.connect(sys.argv[1]).cursor()
rows = [r for r in cur.execute("select name, start, end, grid_x from kernels order by start")]
names = [short(r[0]) for r in rows]
preps = [i for i, n in enumerate(names) if n == "k_prep_lines"]
seg = rows[preps[-1]:]
t0, prev_end, busy = seg[0][1], seg[0][1], 0.0
print("# start_us  dur_us  gap_us  kernel  grid")
for r in seg:
    gap = (r[1] - prev_end) / 1e3
    d = (r[2] - r[1]) / 1e3
    busy += d
    print("%9.1f %8.1f %7.1f  %-30s %d" % ((r[1] - t0) / 1e3, d, gap, short(r[0]), r[3]))
    prev_end = max(prev_end, r[2])
wall = (prev_end - t0) / 1e3
print("# launches %d  wall %.1f us  kernels %.1f us  gaps %.1f us (%.1f %%)" % (len(seg), wall, busy, wall - busy, 100 * (wall - busy) / wall))
