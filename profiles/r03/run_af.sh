#!/bin/bash
set -u
O=$PWD/gpurun_out/r03af; mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_best -- python3 $REPO/profiles/time_scan.py x 100000000 3 best > $O/trace_best.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob
O = "gpurun_out/r03af"
for f in glob.glob("%s/trace_best/**/*kernel_stats.csv" % O, recursive=True):
    tot = 0
    for r in list(csv.DictReader(open(f)))[1:18]:
        print("%-56s calls %5s avg_us %10.2f" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
