#!/usr/bin/env python3
"""Summary of the rocprofv3 --kernel-trace --stats runs of profiles/fastq_shape_bench.py under the three non-DNA modes
(profiles/r05_final_evidence.sh fastq).  Usage: python profiles/fastq_kernel_stats.py <dir with fastq_<mode>/ and fastq_<mode>.log>"""
import csv
import glob
import sys

O = sys.argv[1]
with open(O + "/fastq_kernel_stats.txt", "w") as out:
    for m in ("fail", "convert", "ignore"):
        try:
            line = [l for l in open("%s/fastq_%s.log" % (O, m)) if l.startswith("{")]
        except OSError:
            line = []
        out.write("== shape Q, 25 M four-line FASTQ records (100 M lines, 7.9 GB), --best, non-DNA mode %s (rocprofv3 --kernel-trace --stats) ==\n" % m)
        if line:
            out.write("bench line (under the profiler): " + line[-1][:420] + " ...\n")
        for f in glob.glob("%s/fastq_%s/**/*kernel_stats.csv" % (O, m), recursive=True):
            for r in list(csv.DictReader(open(f)))[:18]:
                if "at::native" in r["Name"] or "rocclr" in r["Name"] or "elementwise" in r["Name"]:
                    continue
                out.write("   %-100s calls %5s avg_us %9.1f total_ms %8.2f\n" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
        out.write("\n")
print(open(O + "/fastq_kernel_stats.txt").read())
