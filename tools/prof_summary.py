#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs tools/prof.sh wrote: per-kernel time stats and per-dispatch mean counters."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


for f in find("kt", "*kernel_stats.csv"):
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("  %-70s calls %6s  avg %10s ns  min %10s  max %10s  pct %s" % (
                row.get("Name", "")[:70], row.get("Calls"), row.get("AverageNs"), row.get("MinNs"), row.get("MaxNs"), row.get("Percentage")))

for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_tcc"):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== counters (%s): mean per dispatch" % sub)
        for k, d in acc.items():
            if "warp_" not in k:
                continue
            print("  " + k)
            for c, v in sorted(d.items()):
                print("     %-24s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
