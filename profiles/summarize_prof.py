#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) into a short text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace", "*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    print("%-60s %8s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in rows[:25]:
        name = r.get("Name", "")[:60]
        print("%-60s %8s %12.1f %12.2f %7s" % (name, r.get("Calls"), float(r.get("TotalDurationNs", 0)) / 1e3,
                                              float(r.get("AverageNs", 0)) / 1e3, r.get("Percentage")))

print()
print("== PMC counters: mean per dispatch, per kernel ==")
agg = defaultdict(lambda: defaultdict(list))
for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_sq2"):
    for f in find(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")[:48]
            agg[k][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
for k in sorted(agg):
    if not any(s in k for s in ("k_forward", "k_nl_", "k_exact", "k_fused", "k_direct", "k_scan", "k_compact")):
        continue
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("    %-28s n=%-5d mean=%.4g" % (c, len(v), sum(v) / len(v)))
