#!/usr/bin/env python3
"""Phase breakdown of the LAST bench.py pass in a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv):
phase A (k_prep_lines + Miller + fexp_a per chunk on the pipelines), the batched inversion, phase C (fexp_c)."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "_mul28" not in r["Kernel_Name"] and "copyBuffer" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
inv = [i for i, r in enumerate(rows) if "k_batch_inv" in r["Kernel_Name"]]
last, prev = inv[-1], inv[-2]
i = prev + 1
while "k_prep_lines" not in rows[i]["Kernel_Name"]:
    i += 1
t0 = int(rows[i]["Start_Timestamp"])
S = lambda r: (int(r["Start_Timestamp"]) - t0) / 1e6
E = lambda r: (int(r["End_Timestamp"]) - t0) / 1e6
seg = rows[i:]
end = max(E(r) for r in seg)
prep = [r for r in seg if "k_prep_lines" in r["Kernel_Name"]]
coop = [r for r in seg if "k_coop" in r["Kernel_Name"]]
print("pass wall %.3f ms" % end)
print("phase A   %.3f ms  (%d k_prep_lines summing %.1f ms, %d k_coop launches summing %.1f ms, two streams overlapped)" % (
    S(rows[last]), len(prep), sum(E(r) - S(r) for r in prep), len(coop) - 1, sum(E(r) - S(r) for r in coop[:-1])))
print("inversion %.3f ms" % (E(rows[last]) - S(rows[last])))
print("phase C   %.3f ms  (one k_coop launch: fexp_c)" % (E(coop[-1]) - S(coop[-1])))
