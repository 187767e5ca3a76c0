#!/usr/bin/env python3
"""Summarise rocprofv3 csv output: per-kernel mean duration from the kernel trace, per-kernel mean of each PMC counter."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 14:
            print(",".join(row[:8]))
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
        acc[k][0] += float(row.get("Counter_Value", 0))
        acc[k][1] += 1
    print("== pmc", os.path.relpath(f, d))
    for (kn, cn), (s, n) in sorted(acc.items()):
        if "wilson" in kn or "stagg" in kn or "cg_" in kn:
            print("  %-60s %-32s mean=%.6g n=%d" % (kn, cn, s / n, n))
