#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) into a short text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
# optional: text bytes scanned by ALL profiled steps together (warm-up included) -> traffic per text byte
text_bytes_total = float(sys.argv[2]) if len(sys.argv) > 2 else None


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

# bench.py scans every placement candidate before its warm-up (k_pair's speed follows the buffer's pages: DESIGN.md section 5), so the
# table above averages the scan kernel over buffers of different speeds.  TIMED_SCAN_DISPATCHES = steps x launches per step: the scan kernel's
# average over the LAST that many dispatches -- the timed steps, all over the chosen buffer: the figure bench.py's roofline.avg_launch_ms is.
nlast = int(os.environ.get("TIMED_SCAN_DISPATCHES", "0"))
if nlast > 0:
    for f in find("trace", "*kernel_trace.csv"):
        rows = [r for r in csv.DictReader(open(f)) if any(s_ in r["Kernel_Name"] for s_ in ("k_pair<", "k_stream<"))]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        last = rows[-nlast:]
        if last:
            us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
            print()
            print("== scan kernel over the timed steps (last %d of %d dispatches: %s) ==" % (len(last), len(rows), last[0]["Kernel_Name"].split("(")[0]))
            print("avg_us %.2f  min %.2f  max %.2f" % (sum(us) / len(us), min(us), max(us)))
            before = rows[:-nlast]
            if before:
                ub = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in before]
                print("the dispatches before them (placement candidates, warm-up): avg_us %.2f  min %.2f  max %.2f" % (sum(ub) / len(ub), min(ub), max(ub)))

print()
print("== PMC counters: mean per dispatch, per kernel ==")
agg = defaultdict(lambda: defaultdict(list))
for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_sq2"):
    for f in find(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")[:48]
            agg[k][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
for k in sorted(agg):
    if not any(s in k for s in ("k_forward", "k_nl_", "k_exact", "k_fused", "k_direct", "k_scan", "k_compact", "k_stream", "k_dfa", "k_pair", "k_packed", "k_multi", "k_verify", "k_emit", "k_nh_top", "k_tiles")):
        continue
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("    %-28s n=%-5d mean=%.4g" % (c, len(v), sum(v) / len(v)))

# Traffic of the dominant scan kernel, priced as MI355X_MICROARCH.md section HBM prescribes:
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide
# (16 B/lane) streaming read, so it is doubled.  The counters sit on the L2's fabric side, so
# Infinity-Cache hits are included.
import json
scan = [k for k in agg if ("k_direct" in k or "k_fused<" in k or "k_forward" in k or "k_stream<" in k or "k_dfa" in k or "k_pair<" in k)
        and "FETCH_SIZE" in agg[k]]
if scan:
    k = max(scan, key=lambda x: sum(agg[x]["FETCH_SIZE"]))
    f, w = agg[k]["FETCH_SIZE"], agg[k].get("WRITE_SIZE", [0.0])
    hbm_total = 2.0 * sum(f) * 1024.0 + (sum(w) * 1024.0 * len(f) / max(1, len(w)))
    res = {"kernel": k.split("(")[0].replace("void ", "").split("<")[0], "dispatches": len(f),
           "fetch_size_kib_mean": sum(f) / len(f), "write_size_kib_mean": sum(w) / max(1, len(w)),
           "hbm_bytes_per_launch_mean": hbm_total / len(f),
           "correction": "2 x FETCH_SIZE (gfx950 wide-read under-count) + WRITE_SIZE, KiB -> bytes"}
    if text_bytes_total:
        res["hbm_bytes_per_text_byte"] = hbm_total / text_bytes_total
    print()
    print("== scan-kernel traffic ==")
    print(json.dumps(res, indent=1))
    with open(os.path.join(out, "pmc_scan_kernel.json"), "w") as fh:
        json.dump(res, fh, indent=1)


# Effective core clock per dispatch of the scan kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration.
# (round 4: the scan kernel runs at two speeds -- is it the clock?)
try:
    dur = {}
    for f in find("pmc_sq2", "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    rows = []
    for f in find("pmc_sq2", "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
                name, ns = dur[r["Dispatch_Id"]]
                if ns > 0 and any(s_ in name for s_ in ("k_pair<", "k_stream<", "k_verify<", "k_exact1<")):
                    rows.append((name.split("(")[0][:40], ns / 1e3, float(r["Counter_Value"]) / 8.0, float(r["Counter_Value"]) / 8.0 / ns))
    if rows:
        print()
        print("== effective clock per dispatch (GRBM_GUI_ACTIVE / 8 / duration) ==")
        by = defaultdict(list)
        for name, us, cyc, ghz in rows:
            by[name].append((us, cyc, ghz))
        for name in sorted(by):
            v = by[name]
            print("%-42s n=%-4d us: min %.1f mean %.1f max %.1f   cycles/8: mean %.4g   GHz: min %.3f mean %.3f max %.3f" % (
                name, len(v), min(x[0] for x in v), sum(x[0] for x in v) / len(v), max(x[0] for x in v),
                sum(x[1] for x in v) / len(v), min(x[2] for x in v), sum(x[2] for x in v) / len(v), max(x[2] for x in v)))
            if "k_pair" in name:
                for us, cyc, ghz in v:
                    print("      %.1f us  %.4g cycles  %.3f GHz" % (us, cyc, ghz))
except Exception as e:   # noqa
    print("clock table: skipped (%s)" % e)
