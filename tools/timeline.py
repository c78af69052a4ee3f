#!/usr/bin/env python3
"""Diagnostic: print one training step's kernel timeline from a rocprofv3 --kernel-trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
adj = [i for i, n in enumerate(names) if "solve_adj" in n]
i0, i1 = adj[-3], adj[-2]
base = int(rows[i0]["End_Timestamp"])
prev_end = base
for r in rows[i0 + 1:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("+%8.1f us  gap %6.1f  dur %7.1f  %s" % ((s - base) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev_end = e
